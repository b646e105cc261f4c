// nyq_imdct.hip -- gfx950 kernels + C ABI (include/nyq_imdct.h) of the batched CELT IMDCT.
//
// The kernels themselves live in nyq_kernels.hpp (lane program: nyq_imdct_lanes.hpp, register
// DFTs: nyq_fft_core.hpp); this file is the host side: context, launch geometry, C ABI.
// HBM traffic per row (shift 0): 3840 B read + 3840 B written (+240 B tail) -- the
// algorithmic 7680 B of SURVEY.md section 8(d); tables (2.4 KB) stay in L2.
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cmath>
#include <cstddef>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <mutex>
#include <new>
#include <string>
#include <vector>

#include "../../include/nyq_imdct.h"

#include "nyq_kernels.hpp"
#include "nyq_post_pipe.hpp"
#include "nyq_chain_kernel.hpp"    // the one-launch frames -> PCM kernel (round 4)
#include "nyq_shape_kernel.hpp"
#include "nyq_entropy_kernel.hpp"    // band shapes from symbol records (round 4)
// The round-1 post-filter kernels (one wave per channel / per stereo pair) and round 2's fused chain are measured-and-
// rejected designs kept for A/B runs: their sources live under tools/ab/ and are compiled only into the tools' build of this
// library (-DNYQ_AB_FORMS -Itools/ab, tools/libnyq_imdct_ab.so), selected through nyq_ctx_set_option; the product has neither.
#ifdef NYQ_AB_FORMS
#include "nyq_chain_fused_r2.hpp"
#endif

using namespace nyq;

using Cfg = DefaultCfg;                 // the shipped kernel configuration (tools/kbench.hip picks it)
constexpr int kWavesPerBlock = Cfg::WPB;

// ---------------------------------------------------------------------------------
// host side of the C ABI
// ---------------------------------------------------------------------------------
struct nyq_ctx {
    int device = 0;
    int cus = 0;
    hipStream_t own_stream = nullptr;
    hipStream_t stream = nullptr;
    float *d_trig = nullptr;     // 481
    float *d_window = nullptr;   // 120
    float h_trig[NYQ_MDCT_N / 4 + 1];
    float h_window[NYQ_OVERLAP];
    // scratch for the host-buffer entry points
    float *d_scratch = nullptr;
    size_t scratch_bytes = 0;
    // copy engines of the host-buffer entry points: uploads and downloads run beside the kernels
    hipStream_t s_h2d = nullptr, s_d2h = nullptr;
    std::vector<hipEvent_t> ev_pool;
    hipEvent_t ev_block = nullptr;       // hipEventBlockingSync: host waits on it sleep instead of spinning
    int res_imdct[4] = {0, 0, 0, 0};   // resident blocks per kernel instance (occupancy query, cached)
    int res_ifft[4] = {0, 0, 0, 0};
    int res_synth_long[4][2] = {{0, 0}, {0, 0}, {0, 0}, {0, 0}};   // [frame size][with the transient-frame role]
    int res_post[4][2] = {{0, 0}, {0, 0}, {0, 0}, {0, 0}};
    int res_post_pipe[4][2] = {{0, 0}, {0, 0}, {0, 0}, {0, 0}};   // [LM][stereo instance]
    int res_chain_fused = 0;             // (A/B build: round 2's fused kernel)
    int res_chain = 0;                   // celt_chain_kernel
    int res_vorbis[12] = {0};
    // options (nyq_ctx_set_option): nothing on a launch path reads the process environment
    int opt_blocks_per_cu = 0;           // 0 = built-in choice
    int opt_post_form = NYQ_POST_FORM_PIPELINE;
    int opt_chain_fused = NYQ_CHAIN_ONE_LAUNCH;
    long opt_chain_window = 0;           // frames per window of the two-kernel chain; 0 = built-in choice
    long opt_host_window = 0;            // frames per time window of the host-buffer frames -> PCM calls; 0 = built-in choice
    int opt_chain_overlap = 0;           // windows: post-filter of window k on a second stream beside the synthesis of window k + 1
    hipStream_t s_post = nullptr;        // (created on first use)
    unsigned *d_pvq = nullptr;           // U(n, k) of the pulse-vector codebooks (shape kernel; created on first use)
    void *d_ent_tables = nullptr;        // the entropy stage's tables (nyq_ctx_set_entropy_tables)
    float *d_vtab = nullptr;             // Vorbis rotation + twiddle tables of every block size, one allocation
    size_t vrot_off[12] = {0}, vtw_off[12] = {0};   // float offsets by log2(n/4)
    std::string err;
    char devname[256];
};

static thread_local std::string g_err;

static int fail(nyq_ctx *ctx, int code, const std::string &msg) {
    if (ctx) ctx->err = msg;
    g_err = msg;
    return code;
}

#define NYQ_HIP(ctx, call)                                                               \
    do {                                                                                 \
        hipError_t e_ = (call);                                                          \
        if (e_ != hipSuccess)                                                            \
            return fail(ctx, NYQ_ERR_HIP, std::string(#call) + ": " + hipGetErrorString(e_)); \
    } while (0)

static void default_tables(float *trig, float *window) {
    // mdct.c:101-102 with mathops.h:83's float PI; modes.c:372-374
    const float PIf = 3.141592653f;
    for (int i = 0; i <= NYQ_MDCT_N / 4; i++) trig[i] = (float)std::cos(2 * PIf * i / NYQ_MDCT_N);
    for (int i = 0; i < NYQ_OVERLAP; i++) {
        double s = std::sin(.5 * M_PI * (i + .5) / NYQ_OVERLAP);
        window[i] = (float)(1.0f * std::sin(.5 * M_PI * s * s));
    }
}

extern "C" const char *nyq_last_error(const nyq_ctx *ctx) { return ctx ? ctx->err.c_str() : g_err.c_str(); }

static int upload_tables(nyq_ctx *ctx) {
    NYQ_HIP(ctx, hipMemcpyAsync(ctx->d_trig, ctx->h_trig, sizeof ctx->h_trig, hipMemcpyHostToDevice, ctx->stream));
    NYQ_HIP(ctx, hipMemcpyAsync(ctx->d_window, ctx->h_window, sizeof ctx->h_window, hipMemcpyHostToDevice, ctx->stream));
    NYQ_HIP(ctx, hipStreamSynchronize(ctx->stream));
    return NYQ_OK;
}

// Vorbis tables (mdct_init, third_party/libvorbis/src/mdct.c:52-91, restated for this factorisation):
// per block size n = 4 << m: rot[i] = (cos, sin)(2 pi (i + 1/8)/n) and twid[k] = (cos, sin)(2 pi k/(n/4)).
static int build_vorbis_tables(nyq_ctx *ctx) {
    std::vector<float> h;
    for (int m = 4; m <= 11; m++) {
        const int n4 = 1 << m, n = 4 * n4;
        ctx->vrot_off[m] = h.size();
        for (int i = 0; i < n4; i++) {
            const double a = 2.0 * M_PI * (i + 0.125) / n;
            h.push_back((float)std::cos(a));
            h.push_back((float)std::sin(a));
        }
        ctx->vtw_off[m] = h.size();
        for (int k = 0; k < n4; k++) {
            const double a = 2.0 * M_PI * k / n4;
            h.push_back((float)std::cos(a));
            h.push_back((float)std::sin(a));
        }
    }
    NYQ_HIP(ctx, hipMalloc(&ctx->d_vtab, h.size() * sizeof(float)));
    NYQ_HIP(ctx, hipMemcpyAsync(ctx->d_vtab, h.data(), h.size() * sizeof(float), hipMemcpyHostToDevice, ctx->stream));
    NYQ_HIP(ctx, hipStreamSynchronize(ctx->stream));
    return NYQ_OK;
}

extern "C" int nyq_device_count(void) {
    int ndev = 0;
    return hipGetDeviceCount(&ndev) == hipSuccess && ndev > 0 ? ndev : 0;
}

extern "C" int nyq_ab_forms_built(void) {
#ifdef NYQ_AB_FORMS
    return 1;
#else
    return 0;
#endif
}

extern "C" int nyq_ctx_set_option(nyq_ctx *ctx, int option, long value) {
    if (!ctx) return fail(nullptr, NYQ_ERR_INVALID, "nyq_ctx_set_option: ctx is NULL");
    switch (option) {
    case NYQ_OPT_BLOCKS_PER_CU:
        if (value < 0 || value > 64) return fail(ctx, NYQ_ERR_INVALID, "nyq_ctx_set_option: NYQ_OPT_BLOCKS_PER_CU must be 0..64");
        if ((int)value != ctx->opt_blocks_per_cu) {            // the cached grid sizes of the row kernels depend on it
            for (int k = 0; k < 4; k++) ctx->res_imdct[k] = ctx->res_ifft[k] = ctx->res_synth_long[k][0] = ctx->res_synth_long[k][1] = 0;
        }
        ctx->opt_blocks_per_cu = (int)value;
        return NYQ_OK;
    case NYQ_OPT_POST_FORM:
        if (value != NYQ_POST_FORM_PIPELINE && value != NYQ_POST_FORM_WAVE_PER_CHANNEL && value != NYQ_POST_FORM_WAVE_PER_PAIR)
            return fail(ctx, NYQ_ERR_INVALID, "nyq_ctx_set_option: unknown NYQ_OPT_POST_FORM value");
        if (value != NYQ_POST_FORM_PIPELINE && !nyq_ab_forms_built())
            return fail(ctx, NYQ_ERR_INVALID, "nyq_ctx_set_option: this build has only the pipeline form (the A/B forms live in tools/libnyq_imdct_ab.so)");
        ctx->opt_post_form = (int)value;
        return NYQ_OK;
    case NYQ_OPT_CHAIN_FUSED:
        if (value != NYQ_CHAIN_TWO_KERNELS && value != NYQ_CHAIN_ONE_LAUNCH && value != NYQ_CHAIN_FUSED_R2)
            return fail(ctx, NYQ_ERR_INVALID, "nyq_ctx_set_option: unknown NYQ_OPT_CHAIN_FUSED value");
        if (value == NYQ_CHAIN_FUSED_R2 && !nyq_ab_forms_built())
            return fail(ctx, NYQ_ERR_INVALID, "nyq_ctx_set_option: round 2's fused chain kernel is an A/B form (tools/libnyq_imdct_ab.so): measured, 3x slower");
        ctx->opt_chain_fused = (int)value;
        return NYQ_OK;
    case NYQ_OPT_CHAIN_OVERLAP:
        if (value != 0 && value != 1) return fail(ctx, NYQ_ERR_INVALID, "nyq_ctx_set_option: NYQ_OPT_CHAIN_OVERLAP must be 0 or 1");
        if (value == 1 && !nyq_ab_forms_built())
            return fail(ctx, NYQ_ERR_INVALID, "nyq_ctx_set_option: the overlapped-window chain is an A/B form (tools/libnyq_imdct_ab.so): measured, not faster");
        ctx->opt_chain_overlap = (int)value;
        return NYQ_OK;
    case NYQ_OPT_HOST_WINDOW:
        if (value < 0 || value % 64 != 0) return fail(ctx, NYQ_ERR_INVALID, "nyq_ctx_set_option: NYQ_OPT_HOST_WINDOW must be a multiple of 64, or 0");
        ctx->opt_host_window = value;
        return NYQ_OK;
    case NYQ_OPT_CHAIN_WINDOW:
        if (value < 0) return fail(ctx, NYQ_ERR_INVALID, "nyq_ctx_set_option: NYQ_OPT_CHAIN_WINDOW must be >= 0");
        ctx->opt_chain_window = value;         // (rounded up to the frame size's chain length at the call)
        return NYQ_OK;
    default:
        return fail(ctx, NYQ_ERR_INVALID, "nyq_ctx_set_option: unknown option");
    }
}

extern "C" int nyq_ctx_get_option(nyq_ctx *ctx, int option, long *value) {
    if (!ctx || !value) return fail(ctx, NYQ_ERR_INVALID, "nyq_ctx_get_option: NULL argument");
    switch (option) {
    case NYQ_OPT_BLOCKS_PER_CU: *value = ctx->opt_blocks_per_cu; return NYQ_OK;
    case NYQ_OPT_POST_FORM: *value = ctx->opt_post_form; return NYQ_OK;
    case NYQ_OPT_CHAIN_FUSED: *value = ctx->opt_chain_fused; return NYQ_OK;
    case NYQ_OPT_CHAIN_WINDOW: *value = ctx->opt_chain_window; return NYQ_OK;
    case NYQ_OPT_CHAIN_OVERLAP: *value = ctx->opt_chain_overlap; return NYQ_OK;
    case NYQ_OPT_HOST_WINDOW: *value = ctx->opt_host_window; return NYQ_OK;
    default: return fail(ctx, NYQ_ERR_INVALID, "nyq_ctx_get_option: unknown option");
    }
}

extern "C" int nyq_ctx_create(nyq_ctx **out, int device) {
    if (!out) return fail(nullptr, NYQ_ERR_INVALID, "nyq_ctx_create: out is NULL");
    *out = nullptr;
    int ndev = 0;
    hipError_t e = hipGetDeviceCount(&ndev);
    if (e != hipSuccess || ndev <= 0)
        return fail(nullptr, NYQ_ERR_NO_DEVICE,
                    std::string("nyq_ctx_create: no HIP device (") + hipGetErrorString(e) +
                        "); this library has no CPU fallback");
    if (device < 0 || device >= ndev)
        return fail(nullptr, NYQ_ERR_NO_DEVICE, "nyq_ctx_create: device index out of range");
    nyq_ctx *ctx = new (std::nothrow) nyq_ctx();
    if (!ctx) return fail(nullptr, NYQ_ERR_ALLOC, "nyq_ctx_create: out of host memory");
    ctx->device = device;
    hipDeviceProp_t prop;
    if ((e = hipSetDevice(device)) != hipSuccess || (e = hipGetDeviceProperties(&prop, device)) != hipSuccess ||
        (e = hipStreamCreateWithFlags(&ctx->own_stream, hipStreamNonBlocking)) != hipSuccess ||
        (e = hipMalloc(&ctx->d_trig, sizeof ctx->h_trig)) != hipSuccess ||
        (e = hipMalloc(&ctx->d_window, sizeof ctx->h_window)) != hipSuccess) {
        std::string m = std::string("nyq_ctx_create: ") + hipGetErrorString(e);
        nyq_ctx_destroy(ctx);
        return fail(nullptr, NYQ_ERR_HIP, m);
    }
    ctx->cus = prop.multiProcessorCount;
    std::snprintf(ctx->devname, sizeof ctx->devname, "%s (%s)", prop.name, prop.gcnArchName);
    ctx->stream = ctx->own_stream;
    default_tables(ctx->h_trig, ctx->h_window);
    int rc = upload_tables(ctx);
    if (rc != NYQ_OK) {
        std::string m = ctx->err;
        nyq_ctx_destroy(ctx);
        return fail(nullptr, rc, m);
    }
    rc = build_vorbis_tables(ctx);
    if (rc != NYQ_OK) {
        std::string m = ctx->err;
        nyq_ctx_destroy(ctx);
        return fail(nullptr, rc, m);
    }
    *out = ctx;
    return NYQ_OK;
}

extern "C" void nyq_ctx_destroy(nyq_ctx *ctx) {
    if (!ctx) return;
    (void)hipSetDevice(ctx->device);
    if (ctx->own_stream) (void)hipStreamSynchronize(ctx->own_stream);
    if (ctx->s_h2d) { (void)hipStreamSynchronize(ctx->s_h2d); (void)hipStreamDestroy(ctx->s_h2d); }
    if (ctx->s_d2h) { (void)hipStreamSynchronize(ctx->s_d2h); (void)hipStreamDestroy(ctx->s_d2h); }
    for (hipEvent_t e : ctx->ev_pool) (void)hipEventDestroy(e);
    if (ctx->ev_block) (void)hipEventDestroy(ctx->ev_block);
    if (ctx->s_post) { (void)hipStreamSynchronize(ctx->s_post); (void)hipStreamDestroy(ctx->s_post); }
    if (ctx->d_scratch) (void)hipFree(ctx->d_scratch);
    if (ctx->d_vtab) (void)hipFree(ctx->d_vtab);
    if (ctx->d_pvq) (void)hipFree(ctx->d_pvq);
    if (ctx->d_ent_tables) (void)hipFree(ctx->d_ent_tables);
    if (ctx->d_trig) (void)hipFree(ctx->d_trig);
    if (ctx->d_window) (void)hipFree(ctx->d_window);
    if (ctx->own_stream) (void)hipStreamDestroy(ctx->own_stream);
    delete ctx;
}

extern "C" int nyq_ctx_set_stream(nyq_ctx *ctx, void *hip_stream) {
    if (!ctx) return fail(nullptr, NYQ_ERR_INVALID, "nyq_ctx_set_stream: ctx is NULL");
    ctx->stream = (hipStream_t)hip_stream;
    return NYQ_OK;
}

extern "C" int nyq_ctx_reset_stream(nyq_ctx *ctx) {
    if (!ctx) return fail(nullptr, NYQ_ERR_INVALID, "nyq_ctx_reset_stream: ctx is NULL");
    ctx->stream = ctx->own_stream;
    return NYQ_OK;
}

extern "C" void *nyq_ctx_get_stream(nyq_ctx *ctx) { return ctx ? (void *)ctx->stream : nullptr; }

extern "C" int nyq_ctx_synchronize(nyq_ctx *ctx) {
    if (!ctx) return fail(nullptr, NYQ_ERR_INVALID, "nyq_ctx_synchronize: ctx is NULL");
    NYQ_HIP(ctx, hipStreamSynchronize(ctx->stream));
    return NYQ_OK;
}

extern "C" int nyq_ctx_set_tables(nyq_ctx *ctx, const float *trig481, const float *window120) {
    if (!ctx || !trig481 || !window120) return fail(ctx, NYQ_ERR_INVALID, "nyq_ctx_set_tables: NULL argument");
    NYQ_HIP(ctx, hipSetDevice(ctx->device));
    NYQ_HIP(ctx, hipStreamSynchronize(ctx->stream));
    std::memcpy(ctx->h_trig, trig481, sizeof ctx->h_trig);
    std::memcpy(ctx->h_window, window120, sizeof ctx->h_window);
    return upload_tables(ctx);
}

extern "C" int nyq_ctx_get_tables(nyq_ctx *ctx, float *trig481, float *window120) {
    if (!ctx || !trig481 || !window120) return fail(ctx, NYQ_ERR_INVALID, "nyq_ctx_get_tables: NULL argument");
    std::memcpy(trig481, ctx->h_trig, sizeof ctx->h_trig);
    std::memcpy(window120, ctx->h_window, sizeof ctx->h_window);
    return NYQ_OK;
}

extern "C" int nyq_ctx_device_info(nyq_ctx *ctx, int *compute_units, char *name, size_t name_len) {
    if (!ctx) return fail(nullptr, NYQ_ERR_INVALID, "nyq_ctx_device_info: ctx is NULL");
    if (compute_units) *compute_units = ctx->cus;
    if (name && name_len) std::snprintf(name, name_len, "%s", ctx->devname);
    return NYQ_OK;
}

static bool aligned16(const void *p) { return (reinterpret_cast<uintptr_t>(p) & 15u) == 0; }

// persistent grid: never more blocks than the chip keeps resident (registers, LDS and wave
// slots all counted by the occupancy query) -- a larger grid would run its surplus blocks as a
// second, mostly idle round -- capped at the measured sweet spot of NYQ_WAVES_PER_CU, and never
// more than the work.
// waves_per_cu: the measured sweet spot of the kernel at hand (tools/bpc_ab.py, one process, interleaved, MI355X):
//   rows of 3840 / 1920 B (15 KB of loads in flight per wave-group): 6 -- more streams lower HBM efficiency;
//   rows of 960 / 480 B (7.7 KB per wave-group): 8 / 7 for frame synthesis (LM 1: 1.043 -> 0.989 ms, LM 0: 0.856 -> 0.825),
//   7-8 for the plain row kernel of 1920 / 960 / 480 B rows (2-4 %).
template <typename K>
static int resident_blocks(nyq_ctx *ctx, K kernel, int *cache, int waves_per_cu = NYQ_WAVES_PER_CU) {
    if (*cache > 0) return *cache;
    int per_cu = 0;
    hipError_t e = hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, kernel, kWave * kWavesPerBlock, 0);
    if (e != hipSuccess || per_cu < 1) per_cu = 1;
    const int want = (waves_per_cu + kWavesPerBlock - 1) / kWavesPerBlock;   // see nyq_kernels.hpp
    if (per_cu > want) per_cu = want;
    if (ctx->opt_blocks_per_cu > 0) per_cu = ctx->opt_blocks_per_cu;   // NYQ_OPT_BLOCKS_PER_CU: tuning knob for profiling runs
    *cache = per_cu * ctx->cus;
    return *cache;
}

static unsigned grid_for(size_t batch, int rows_per_group, int resident) {
    const size_t ngroups = (batch + rows_per_group - 1) / rows_per_group;
    const size_t need = (ngroups + kWavesPerBlock - 1) / kWavesPerBlock;
    return (unsigned)(need < (size_t)resident ? need : (size_t)resident);
}

template <int N2R>
static int launch_imdct(nyq_ctx *ctx, const float *d_in, const float *d_carry, float *d_fin, float *d_tail,
                        size_t batch) {
    const int res = resident_blocks(ctx, imdct_rows_kernel<N2R, Cfg>, &ctx->res_imdct[Geo<N2R>::SHIFT],
                                    N2R == 32 ? 6 : N2R == 8 ? 8 : 7);
    hipLaunchKernelGGL((imdct_rows_kernel<N2R, Cfg>), dim3(grid_for(batch, Geo<N2R>::G, res)), dim3(kWave * kWavesPerBlock), 0,
                       ctx->stream, d_in, d_carry, d_fin, d_tail, (long)batch, ctx->d_trig, ctx->d_window);
    NYQ_HIP(ctx, hipGetLastError());
    return NYQ_OK;
}

template <int N2R>
static int launch_ifft(nyq_ctx *ctx, const float *d_in, float *d_out, size_t batch) {
    const int res = resident_blocks(ctx, ifft_rows_kernel<N2R, kWavesPerBlock>, &ctx->res_ifft[Geo<N2R>::SHIFT]);
    hipLaunchKernelGGL((ifft_rows_kernel<N2R, kWavesPerBlock>), dim3(grid_for(batch, Geo<N2R>::G, res)), dim3(kWave * kWavesPerBlock), 0,
                       ctx->stream, d_in, d_out, (long)batch);
    NYQ_HIP(ctx, hipGetLastError());
    return NYQ_OK;
}

extern "C" int nyq_imdct_batch_dev(nyq_ctx *ctx, int shift, const float *d_in, const float *d_carry, float *d_fin,
                                   float *d_tail, size_t batch) {
    if (!ctx) return fail(nullptr, NYQ_ERR_INVALID, "nyq_imdct_batch_dev: ctx is NULL");
    if (shift < 0 || shift > 3) return fail(ctx, NYQ_ERR_INVALID, "nyq_imdct_batch_dev: shift must be 0..3");
    if (batch == 0) return NYQ_OK;
    if (!d_in || !d_fin) return fail(ctx, NYQ_ERR_INVALID, "nyq_imdct_batch_dev: NULL in/fin");
    if (!aligned16(d_in) || !aligned16(d_fin) || !aligned16(d_carry) || !aligned16(d_tail))
        return fail(ctx, NYQ_ERR_INVALID, "nyq_imdct_batch_dev: device pointers must be 16-byte aligned");
    switch (shift) {
    case 0: return launch_imdct<32>(ctx, d_in, d_carry, d_fin, d_tail, batch);
    case 1: return launch_imdct<16>(ctx, d_in, d_carry, d_fin, d_tail, batch);
    case 2: return launch_imdct<8>(ctx, d_in, d_carry, d_fin, d_tail, batch);
    default: return launch_imdct<4>(ctx, d_in, d_carry, d_fin, d_tail, batch);
    }
}

extern "C" int nyq_ifft_batch_dev(nyq_ctx *ctx, int nfft, const float *d_in, float *d_out, size_t batch) {
    if (!ctx) return fail(nullptr, NYQ_ERR_INVALID, "nyq_ifft_batch_dev: ctx is NULL");
    if (nfft != 480 && nfft != 240 && nfft != 120 && nfft != 60)
        return fail(ctx, NYQ_ERR_INVALID, "nyq_ifft_batch_dev: nfft must be 480, 240, 120 or 60");
    if (batch == 0) return NYQ_OK;
    if (!d_in || !d_out || d_in == d_out)
        return fail(ctx, NYQ_ERR_INVALID, "nyq_ifft_batch_dev: NULL or in-place buffers (kiss_fft.c:708: in-place not supported)");
    if (!aligned16(d_in) || !aligned16(d_out))
        return fail(ctx, NYQ_ERR_INVALID, "nyq_ifft_batch_dev: device pointers must be 16-byte aligned");
    switch (nfft) {
    case 480: return launch_ifft<32>(ctx, d_in, d_out, batch);
    case 240: return launch_ifft<16>(ctx, d_in, d_out, batch);
    case 120: return launch_ifft<8>(ctx, d_in, d_out, batch);
    default: return launch_ifft<4>(ctx, d_in, d_out, batch);
    }
}

// ---- frame sequences ----------------------------------------------------------------
extern "C" size_t nyq_celt_synth_work_floats(size_t nstreams, size_t nframes, int channels) {
    return nstreams * (size_t)(channels > 0 ? channels : 0) * (nframes + 1) * NYQ_HALF_OV;
}

// ONE launch for the long frames and the transient frames of a call (synth_frames_kernel): workgroups [0, nlong) take the
// long frames -- the persistent grid of the size's measured sweet spot --, kShortWavesPerCU more per CU the transient ones
// (they are a few per cent of the frames and finish well inside the long frames' span at one wave per CU: measured with
// 1, 2, 3, 4 and 6, the call took the same time to 0.4 %).  ONE, not more: the 960- and 480-sample instances take 205 / 175
// VGPRs (a cap of 168 makes them spill), i.e. eight waves fit a CU, six are the long frames' -- a seventh leaves a slot of
// slack, an eighth would be the exact fit that places long-lived workgroups late (nyq_post_pipe.hpp).
constexpr int kShortWavesPerCU = 1;
template <int N2R, int LMc>
static int launch_synth_frames(nyq_ctx *ctx, const SynthArgs &A, int *cache) {
    const size_t nchunks = (size_t)A.nstreams * A.channels * FrameLongRows<N2R>::chunks_per_channel(A.nframes);
    const int res = resident_blocks(ctx, synth_frames_kernel<N2R, LMc, Cfg>, cache, N2R >= 16 ? 6 : N2R == 8 ? 8 : 7);
    const size_t need = (nchunks + kWavesPerBlock - 1) / kWavesPerBlock;
    const unsigned nlong = (unsigned)(need < (size_t)res ? need : (size_t)res);
    unsigned nshort = 0;
    if (LMc > 0) {
        const size_t units = (size_t)A.nstreams * A.channels * A.nframes;
        const size_t want = (units + kWave * kWavesPerBlock - 1) / (kWave * kWavesPerBlock);
        const size_t cap = (size_t)ctx->cus * ((kShortWavesPerCU + kWavesPerBlock - 1) / kWavesPerBlock);
        nshort = (unsigned)(want < cap ? want : cap);
    }
    hipLaunchKernelGGL((synth_frames_kernel<N2R, LMc, Cfg>), dim3(nlong + nshort), dim3(kWave * kWavesPerBlock), 0, ctx->stream, A,
                       ctx->d_trig, ctx->d_window, (int)nlong);
    NYQ_HIP(ctx, hipGetLastError());
    return NYQ_OK;
}

// shared by nyq_celt_synth_dev and nyq_imdct_chain_dev (a chain is a 1-channel stream without
// transient frames); arguments already validated
static int synth_core(nyq_ctx *ctx, int LM, const float *d_freq, const unsigned char *d_transient, float *d_pcm,
                      const float *d_state_in, float *d_state_out, float *d_work, size_t nstreams, size_t nframes,
                      int channels, size_t fstride = 0) {
    const size_t nsc = nstreams * (size_t)channels;
    // the fix-up pass launches a (stream * channel) x (frames / 256) grid: 2^31 - 1 units, 16.7 M frames per call
    if (nsc > 0x7fffffffu || (nframes + 255) / 256 > 65535)
        return fail(ctx, NYQ_ERR_INVALID, "frame synthesis: more than 2^31 - 1 (stream, channel) units or 16.7 M frames in one call");
    // (slot 0 of every (stream, channel)'s tails row is unused: the fix-up pass reads the state handed in instead.  Round 2
    // kept a memset node here because the post-filter kernel behind the synthesis ran 0.9 instead of 1.6 ms with it; the
    // cause turned out to be workgroup placement of that kernel -- nyq_post_pipe.hpp, kPipeHist -- and is fixed there.)
    SynthArgs A;
    A.freq = d_freq;
    A.transient = LM > 0 ? d_transient : nullptr;   // LM 0: one block either way (B = 1)
    A.pcm = d_pcm;
    A.tails = d_work;
    A.nstreams = (long)nstreams;
    A.nframes = (long)nframes;
    A.channels = channels;
    A.state_in = d_state_in;                        // (read and replaced by the fix-up pass: no copy launches)
    A.state_out = d_state_out;
    A.fstride = (long)fstride;                      // 0: freq / transient are dense
    int rc, chain_frames;
    // [LM][with transient frames]: the instance without the transient role is what chains and flag-less calls launch
    const bool tr = A.transient != nullptr;
    switch (LM) {
    case 3: rc = tr ? launch_synth_frames<32, 3>(ctx, A, &ctx->res_synth_long[0][1]) : launch_synth_frames<32, 0>(ctx, A, &ctx->res_synth_long[0][0]);
            chain_frames = Geo<32>::CHAIN_FRAMES; break;
    case 2: rc = tr ? launch_synth_frames<16, 2>(ctx, A, &ctx->res_synth_long[1][1]) : launch_synth_frames<16, 0>(ctx, A, &ctx->res_synth_long[1][0]);
            chain_frames = Geo<16>::CHAIN_FRAMES; break;
    case 1: rc = tr ? launch_synth_frames<8, 1>(ctx, A, &ctx->res_synth_long[2][1]) : launch_synth_frames<8, 0>(ctx, A, &ctx->res_synth_long[2][0]);
            chain_frames = Geo<8>::CHAIN_FRAMES; break;
    default: rc = launch_synth_frames<4, 0>(ctx, A, &ctx->res_synth_long[3][0]); chain_frames = Geo<4>::CHAIN_FRAMES; break;
    }
    if (rc != NYQ_OK) return rc;
    const size_t per_block = (size_t)kWave * kFixupWaves;
    // (grid.x = (stream, channel) units, grid.y = blocks of 256 frames: the limits are checked on entry)
    hipLaunchKernelGGL(synth_fixup_kernel, dim3((unsigned)nsc, (unsigned)((nframes + per_block - 1) / per_block)),
                       dim3(kWave * kFixupWaves), 0, ctx->stream, A, 120 << LM, chain_frames, ctx->d_window);
    NYQ_HIP(ctx, hipGetLastError());
    return NYQ_OK;
}

extern "C" int nyq_celt_synth_dev(nyq_ctx *ctx, int LM, const float *d_freq, const unsigned char *d_transient,
                                  float *d_pcm, float *d_state, float *d_work, size_t nstreams, size_t nframes,
                                  int channels) {
    if (!ctx) return fail(nullptr, NYQ_ERR_INVALID, "nyq_celt_synth_dev: ctx is NULL");
    if (LM < 0 || LM > 3) return fail(ctx, NYQ_ERR_INVALID, "nyq_celt_synth_dev: LM must be 0..3");
    if (channels < 1 || channels > 255) return fail(ctx, NYQ_ERR_INVALID, "nyq_celt_synth_dev: channels must be 1..255");
    if (nstreams == 0 || nframes == 0) return NYQ_OK;
    if (!d_freq || !d_pcm || !d_work) return fail(ctx, NYQ_ERR_INVALID, "nyq_celt_synth_dev: NULL freq/pcm/work");
    if (!aligned16(d_freq) || !aligned16(d_pcm) || !aligned16(d_work) || !aligned16(d_state))
        return fail(ctx, NYQ_ERR_INVALID, "nyq_celt_synth_dev: device pointers must be 16-byte aligned");
    return synth_core(ctx, LM, d_freq, d_transient, d_pcm, d_state, d_state, d_work, nstreams, nframes, channels);
}

extern "C" int nyq_imdct_chain_dev(nyq_ctx *ctx, int shift, const float *d_in, const float *d_carry0, float *d_pcm,
                                   float *d_tail_out, float *d_work, size_t nchains, size_t len) {
    if (!ctx) return fail(nullptr, NYQ_ERR_INVALID, "nyq_imdct_chain_dev: ctx is NULL");
    if (shift < 0 || shift > 3) return fail(ctx, NYQ_ERR_INVALID, "nyq_imdct_chain_dev: shift must be 0..3");
    if (nchains * len == 0) return NYQ_OK;
    if (!d_in || !d_pcm || !d_work) return fail(ctx, NYQ_ERR_INVALID, "nyq_imdct_chain_dev: NULL in/pcm/work");
    if (!aligned16(d_in) || !aligned16(d_pcm) || !aligned16(d_work) || !aligned16(d_carry0) || !aligned16(d_tail_out))
        return fail(ctx, NYQ_ERR_INVALID, "nyq_imdct_chain_dev: device pointers must be 16-byte aligned");
    // a chain is a one-channel stream of `len` long frames of size N2 = 120 << (3 - shift)
    return synth_core(ctx, 3 - shift, d_in, nullptr, d_pcm, d_carry0, d_tail_out, d_work, nchains, len, 1);
}

// ---- post-filter + de-emphasis + interleave ---------------------------------------------
#ifdef NYQ_AB_FORMS
#include "nyq_post_kernels.hpp"
// round-1 forms (A/B build only)
template <int LM, int NC>
static int launch_post(nyq_ctx *ctx, const PostArgs &A) {
    // 12 KB of LDS per channel a wave owns: two waves per block when a wave owns two channels, so that three
    // blocks (six waves) fit a CU either way
    constexpr int kPostWavesPerBlock = NC == 2 ? 2 : 4;
    const size_t nunits = (size_t)A.nstreams * (size_t)(A.channels / NC);
    int &res = ctx->res_post[LM][NC - 1];
    if (res == 0) {
        int per_cu = 0;
        hipError_t e = hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, celt_post_kernel<LM, kPostWavesPerBlock, NC>,
                                                                    kWave * kPostWavesPerBlock, 0);
        if (e != hipSuccess || per_cu < 1) per_cu = 1;
        // one workgroup per CU fewer than what fits on paper, four at most (the design point: 8 chains per CU).  A
        // workgroup lives for the whole launch, and "exactly what fits" only fits when the CU's LDS and register
        // allocators start from zero -- see kPipeHist in nyq_post_pipe.hpp; 1280 stereo streams at five per CU: 1.75 ms,
        // two rounds for some CUs, where 1024 take 1.0
        per_cu = per_cu > 4 ? 4 : per_cu > 1 ? per_cu - 1 : 1;
        res = per_cu * ctx->cus;
    }
    const size_t need = (nunits + kPostWavesPerBlock - 1) / kPostWavesPerBlock;
    const unsigned grid = (unsigned)(need < (size_t)res ? need : (size_t)res);
    hipLaunchKernelGGL((celt_post_kernel<LM, kPostWavesPerBlock, NC>), dim3(grid), dim3(kWave * kPostWavesPerBlock), 0,
                       ctx->stream, A, ctx->d_window);
    NYQ_HIP(ctx, hipGetLastError());
    return NYQ_OK;
}

#endif   // NYQ_AB_FORMS

// the workgroup-pipelined form (nyq_post_pipe.hpp): one workgroup of 2 comb waves + 1 I/O wave per two chains
constexpr int kPipeGroupsPerCU = 4;      // workgroups kept per CU: one ROUND of the stage is 4 x CUs workgroups = 8 x CUs chains
extern "C" size_t nyq_celt_post_round_chains(nyq_ctx *ctx) {
    return ctx ? (size_t)kPipeGroupsPerCU * (size_t)kPipeUnits * (size_t)ctx->cus : 0;
}

template <int LM>
static int launch_post_pipe(nyq_ctx *ctx, const PostArgs &A) {
    const size_t nunits = (size_t)A.nstreams * (size_t)A.channels;
    const size_t npairs = (nunits + kPipeUnits - 1) / kPipeUnits;
    // the launched instance's own occupancy: <LM, true> for stereo streams, <LM, false> otherwise
    const bool pair = A.channels == 2;
    int &res = ctx->res_post_pipe[LM][pair ? 1 : 0];
    if (res == 0) {
        int per_cu = 0;
        hipError_t e = pair ? hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, celt_post_pipe_kernel<LM, true>, kWave * kPipeWaves, 0)
                            : hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, celt_post_pipe_kernel<LM, false>, kWave * kPipeWaves, 0);
        if (e != hipSuccess || per_cu < 1) per_cu = 1;
        // one workgroup per CU fewer than what fits on paper, kPipeGroupsPerCU (four) at most: the design point is 8 chains
        // per CU.  A workgroup lives for the whole launch, and "exactly what fits" only fits when the CU's LDS and register
        // allocators start from zero -- see kPipeHist in nyq_post_pipe.hpp; 1280 stereo streams at five per CU: 1.75 ms,
        // two rounds for some CUs, where 1024 take 1.0 (the round size of this stage)
        per_cu = per_cu > kPipeGroupsPerCU ? kPipeGroupsPerCU : per_cu > 1 ? per_cu - 1 : 1;
        res = per_cu * ctx->cus;
    }
    const unsigned grid = (unsigned)(npairs < (size_t)res ? npairs : (size_t)res);
    if (pair)
        hipLaunchKernelGGL((celt_post_pipe_kernel<LM, true>), dim3(grid), dim3(kWave * kPipeWaves), 0, ctx->stream, A, ctx->d_window);
    else
        hipLaunchKernelGGL((celt_post_pipe_kernel<LM, false>), dim3(grid), dim3(kWave * kPipeWaves), 0, ctx->stream, A, ctx->d_window);
    NYQ_HIP(ctx, hipGetLastError());
    return NYQ_OK;
}

static int post_core(nyq_ctx *ctx, int LM, const float *d_pcm, const int *d_pf_pitch, const float *d_pf_gain,
                     const int *d_pf_tapset, const float *d_pf_state_in, float *d_pf_state_out, float *d_hist, float *d_deemph,
                     float *d_out, size_t nstreams, size_t nframes, int channels, size_t pstride, const OutDesc *d_desc = nullptr) {
    PostArgs A;
    A.pcm = d_pcm;
    A.pf_pitch = d_pf_pitch;
    A.pf_gain = d_pf_gain;
    A.pf_tapset = d_pf_tapset;
    A.pf_state = d_pf_state_in;
    A.pf_state_out = d_pf_state_out;
    A.hist = d_hist;
    A.deemph = d_deemph;
    A.out = d_out;
    A.nstreams = (long)nstreams;
    A.nframes = (long)nframes;
    A.channels = channels;
    A.pstride = (long)pstride;                      // 0: pf_* / out are dense
    A.desc = d_desc;                                // per-stream destinations in a file's interleaved layout, or null
    // The workgroup pipeline (a chain's wave issues only the recursion) is the product's one form.  The
    // round-1 forms -- one wave per channel, or per stereo pair, doing everything -- exist in the A/B build only and are
    // chosen per context: nyq_ctx_set_option(ctx, NYQ_OPT_POST_FORM, ...).
#ifdef NYQ_AB_FORMS
    if (ctx->opt_post_form == NYQ_POST_FORM_WAVE_PER_PAIR && channels == 2) {
        switch (LM) {
            case 0: return launch_post<0, 2>(ctx, A);
            case 1: return launch_post<1, 2>(ctx, A);
            case 2: return launch_post<2, 2>(ctx, A);
            default: return launch_post<3, 2>(ctx, A);
        }
    }
    if (ctx->opt_post_form != NYQ_POST_FORM_PIPELINE) {
        switch (LM) {
            case 0: return launch_post<0, 1>(ctx, A);
            case 1: return launch_post<1, 1>(ctx, A);
            case 2: return launch_post<2, 1>(ctx, A);
            default: return launch_post<3, 1>(ctx, A);
        }
    }
#endif
    switch (LM) {
        case 0: return launch_post_pipe<0>(ctx, A);
        case 1: return launch_post_pipe<1>(ctx, A);
        case 2: return launch_post_pipe<2>(ctx, A);
        default: return launch_post_pipe<3>(ctx, A);
    }
}

extern "C" int nyq_celt_post_dev(nyq_ctx *ctx, int LM, const float *d_pcm, const int *d_pf_pitch, const float *d_pf_gain,
                                 const int *d_pf_tapset, const float *d_pf_state_in, float *d_pf_state_out,
                                 float *d_hist, float *d_deemph, float *d_out, size_t nstreams, size_t nframes,
                                 int channels) {
    if (!ctx) return fail(nullptr, NYQ_ERR_INVALID, "nyq_celt_post_dev: ctx is NULL");
    if (LM < 0 || LM > 3) return fail(ctx, NYQ_ERR_INVALID, "nyq_celt_post_dev: LM must be 0..3");
    if (channels < 1 || channels > 255) return fail(ctx, NYQ_ERR_INVALID, "nyq_celt_post_dev: channels must be 1..255");
    if (nstreams == 0 || nframes == 0) return NYQ_OK;
    if (!d_pcm || !d_pf_pitch || !d_pf_gain || !d_pf_tapset || !d_out)
        return fail(ctx, NYQ_ERR_INVALID, "nyq_celt_post_dev: NULL pcm/parameters/out");
    if (d_pf_state_in && d_pf_state_in == d_pf_state_out)
        return fail(ctx, NYQ_ERR_INVALID, "nyq_celt_post_dev: pf_state_in and pf_state_out must not alias");
    if (((uintptr_t)d_pcm | (uintptr_t)d_out) & 15)
        return fail(ctx, NYQ_ERR_INVALID, "nyq_celt_post_dev: pcm and out must be 16-byte aligned");
    return post_core(ctx, LM, d_pcm, d_pf_pitch, d_pf_gain, d_pf_tapset, d_pf_state_in, d_pf_state_out, d_hist, d_deemph, d_out,
                     nstreams, nframes, channels, 0);
}

// ---- freq[] -> PCM: the chain as one operator ------------------------------------------------------
extern "C" int nyq_celt_chain_fused_supported(int LM, int channels) { return LM == 3 && channels == 2; }

static size_t round16f(size_t nfloats);

// frames per time window of the two-kernel chain when the caller has not chosen (NYQ_OPT_CHAIN_WINDOW = 0)
static size_t default_chain_window(size_t nstreams, size_t nframes, int channels, int LM) {
    (void)nstreams; (void)channels; (void)LM;
    return nframes;                                 // one window: see DESIGN.md 4.8 for the measurements behind it
}

// fstride: 0 = dense; otherwise freq / transient / pf_* / out are windows of per-stream arrays `fstride` frames long
static int chain_core(nyq_ctx *ctx, int LM, const float *d_freq, const unsigned char *d_transient,
                      const int *d_pf_pitch, const float *d_pf_gain, const int *d_pf_tapset,
                      const float *d_pf_state_in, float *d_pf_state_out, float *d_overlap, float *d_hist,
                      float *d_deemph, float *d_out, float *d_pcm, float *d_work, size_t nstreams,
                      size_t nframes, int channels, size_t fstride, const OutDesc *d_desc = nullptr) {
    if (!ctx) return fail(nullptr, NYQ_ERR_INVALID, "nyq_celt_chain_dev: ctx is NULL");
    if (d_desc && channels > 2) return fail(ctx, NYQ_ERR_INVALID, "nyq_celt_chain_mapped_dev: output descriptors serve mono and stereo streams");
    if (LM < 0 || LM > 3) return fail(ctx, NYQ_ERR_INVALID, "nyq_celt_chain_dev: LM must be 0..3");
    if (channels < 1 || channels > 255) return fail(ctx, NYQ_ERR_INVALID, "nyq_celt_chain_dev: channels must be 1..255");
    if (nstreams == 0 || nframes == 0) return NYQ_OK;
    if (!d_freq || !d_pf_pitch || !d_pf_gain || !d_pf_tapset || !d_out)
        return fail(ctx, NYQ_ERR_INVALID, "nyq_celt_chain_dev: NULL freq/parameters/out");
    if (d_pf_state_in && d_pf_state_in == d_pf_state_out)
        return fail(ctx, NYQ_ERR_INVALID, "nyq_celt_chain_dev: pf_state_in and pf_state_out must not alias");
    if (!aligned16(d_freq) || !aligned16(d_out) || !aligned16(d_overlap) || !aligned16(d_hist))
        return fail(ctx, NYQ_ERR_INVALID, "nyq_celt_chain_dev: device pointers must be 16-byte aligned");
    // The one-launch kernel (nyq_chain_kernel.hpp): the post-filter's workgroup plus a transform wave that works in place in
    // the frame regions -- the time-domain frame never leaves the CU.
    if (ctx->opt_chain_fused == NYQ_CHAIN_ONE_LAUNCH && nyq_celt_chain_fused_supported(LM, channels)) {
        ChainArgs A;
        A.freq = d_freq;
        A.transient = d_transient;
        A.ov_state = d_overlap;
        A.pf_pitch = d_pf_pitch;
        A.pf_gain = d_pf_gain;
        A.pf_tapset = d_pf_tapset;
        A.pf_state = d_pf_state_in;
        A.pf_state_out = d_pf_state_out;
        A.hist = d_hist;
        A.deemph = d_deemph;
        A.out = d_out;
        A.nstreams = (long)nstreams;
        A.nframes = (long)nframes;
        A.fstride = (long)fstride;
        A.pstride = (long)fstride;
        A.desc = d_desc;
        if (ctx->res_chain == 0) {
            int per_cu = 0;
            hipError_t e = hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, celt_chain_kernel, kWave * kChainWaves, 0);
            if (e != hipSuccess || per_cu < 1) per_cu = 1;
            // four workgroups = 8 chains per CU, as the post-filter pipeline (nyq_celt_post_round_chains)
            ctx->res_chain = (per_cu > kPipeGroupsPerCU ? kPipeGroupsPerCU : per_cu) * ctx->cus;
        }
        const unsigned grid = (unsigned)(nstreams < (size_t)ctx->res_chain ? nstreams : (size_t)ctx->res_chain);
        hipLaunchKernelGGL(celt_chain_kernel, dim3(grid), dim3(kWave * kChainWaves), 0, ctx->stream, A, ctx->d_trig, ctx->d_window);
        NYQ_HIP(ctx, hipGetLastError());
        return NYQ_OK;
    }
#ifdef NYQ_AB_FORMS
    // Round 2's fused kernel (A/B build only) is correct (tests/test_gpu_chain.py) and measured 3x SLOWER than the two launches
    // (6.1 ms vs 1.9 ms for 1024 x 256 stereo frames, profiles/r02_*): one IMDCT wave per four chains cannot keep up with
    // the comb waves, and the LDS that would hold more IMDCT slices is what keeps every chain resident.
    if (ctx->opt_chain_fused == NYQ_CHAIN_FUSED_R2 && nyq_celt_chain_fused_supported(LM, channels) && fstride == 0 && !d_desc) {
        FusedR2Args A;
        A.freq = d_freq;
        A.transient = d_transient;
        A.ov_state = d_overlap;
        A.pf_pitch = d_pf_pitch;
        A.pf_gain = d_pf_gain;
        A.pf_tapset = d_pf_tapset;
        A.pf_state = d_pf_state_in;
        A.pf_state_out = d_pf_state_out;
        A.hist = d_hist;
        A.deemph = d_deemph;
        A.out = d_out;
        A.nstreams = (long)nstreams;
        A.nframes = (long)nframes;
        if (ctx->res_chain_fused == 0) {
            int per_cu = 0;
            hipError_t e = hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, celt_chain_fused_kernel, kWave * kFuseWaves, 0);
            if (e != hipSuccess || per_cu < 1) per_cu = 1;
            ctx->res_chain_fused = per_cu * ctx->cus;
        }
        const size_t ngroups = (nstreams + 1) / 2;
        const unsigned grid = (unsigned)(ngroups < (size_t)ctx->res_chain_fused ? ngroups : (size_t)ctx->res_chain_fused);
        hipLaunchKernelGGL(celt_chain_fused_kernel, dim3(grid), dim3(kWave * kFuseWaves), 0, ctx->stream, A, ctx->d_trig, ctx->d_window);
        NYQ_HIP(ctx, hipGetLastError());
        return NYQ_OK;
    }
#endif
    if (!d_pcm || !d_work)
        return fail(ctx, NYQ_ERR_INVALID, "nyq_celt_chain_dev: the two-kernel chain needs d_pcm and d_work");
    if (!aligned16(d_pcm) || !aligned16(d_work))
        return fail(ctx, NYQ_ERR_INVALID, "nyq_celt_chain_dev: device pointers must be 16-byte aligned");
    // Time windows (NYQ_OPT_CHAIN_WINDOW): synthesis and post-filter alternate over windows of W frames (a multiple of
    // 64: the synthesis kernels' in-wave carry chains then restart at the same frames as in one call, so the result is
    // bit-identical).  A window's time-domain frames live in the FIRST nsc * W * N floats of d_pcm, written by one launch
    // and read by the next: sized to the 256 MB Infinity Cache they need not make the round trip through HBM that the
    // whole-call form pays (DESIGN.md 4.8 for what that was measured to be worth).  The decoder states travel from window to
    // window through the caller's state buffers, or -- where the caller passed NULL (fresh decoder, state discarded) --
    // through the part of d_work that a window's synthesis does not use.
    const size_t N = (size_t)120 << LM, nsc = nstreams * (size_t)channels;
    size_t W = ctx->opt_chain_window > 0 ? (size_t)ctx->opt_chain_window : default_chain_window(nstreams, nframes, channels, LM);
    const size_t chainlen = LM == 3 ? Geo<32>::CHAIN_FRAMES : LM == 2 ? Geo<16>::CHAIN_FRAMES : LM == 1 ? Geo<8>::CHAIN_FRAMES : Geo<4>::CHAIN_FRAMES;
    W = (W + chainlen - 1) / chainlen * chainlen;   // windows start where the in-wave carry chains start: bit-identical to one window
    if (W >= nframes || nframes - W < 20 || fstride != 0 || d_desc) {   // (the temporaries below need 20 frames' worth of d_work beyond a window's own)
        int rc = synth_core(ctx, LM, d_freq, d_transient, d_pcm, d_overlap, d_overlap, d_work, nstreams, nframes, channels, fstride);
        if (rc != NYQ_OK) return rc;
        return post_core(ctx, LM, d_pcm, d_pf_pitch, d_pf_gain, d_pf_tapset, d_pf_state_in, d_pf_state_out, d_hist, d_deemph,
                         d_out, nstreams, nframes, channels, fstride, d_desc);
    }
    // temporaries behind the window's own share of d_work (nsc * (W + 1) * 60 floats of nsc * (nframes + 1) * 60):
    // overlap [nsc][60], hist [nsc][1088], deemph [nsc], post-filter state ping / pong [nstreams][6] -- 1150 floats per
    // (stream, channel) at most, and nframes - W >= 64 frames' worth (3840 floats each) are free
    float *t_ov = d_work + round16f(nsc * (W + 1) * NYQ_HALF_OV), *t_hi = t_ov + round16f(nsc * NYQ_HALF_OV),
          *t_de = t_hi + round16f(nsc * kPostHist), *t_pf[2] = {t_de + round16f(nsc), t_de + round16f(nsc) + round16f(nstreams * 6)};
    float *ov = d_overlap ? d_overlap : t_ov, *hi = d_hist ? d_hist : t_hi, *de = d_deemph ? d_deemph : t_de;
    if (!d_overlap) NYQ_HIP(ctx, hipMemsetAsync(t_ov, 0, nsc * NYQ_HALF_OV * sizeof(float), ctx->stream));
    if (!d_hist) NYQ_HIP(ctx, hipMemsetAsync(t_hi, 0, nsc * kPostHist * sizeof(float), ctx->stream));
    if (!d_deemph) NYQ_HIP(ctx, hipMemsetAsync(t_de, 0, nsc * sizeof(float), ctx->stream));
    const float *pf_in = d_pf_state_in;
    // NYQ_OPT_CHAIN_OVERLAP: the post-filter of window k runs on a second stream beside the synthesis of window k + 1 (two
    // window buffers of d_pcm and d_work alternate).  An experiment, measured and not the default: both
    // kernels stream at the memory system's rate and size their grids for the whole chip, so they take turns.
    const size_t nwin = (nframes + W - 1) / W;
    const size_t wfl = round16f(nsc * (W + 1) * NYQ_HALF_OV), pfl = round16f(nsc * W * N);
    const bool overlap = ctx->opt_chain_overlap && nwin >= 2 && nframes >= 2 * W + 40;
    if (overlap) {
        if (!ctx->s_post) NYQ_HIP(ctx, hipStreamCreateWithFlags(&ctx->s_post, hipStreamNonBlocking));
        while (ctx->ev_pool.size() < 2 * nwin + 1) {
            hipEvent_t e;
            NYQ_HIP(ctx, hipEventCreateWithFlags(&e, hipEventDisableTiming));
            ctx->ev_pool.push_back(e);
        }
        // temporaries behind BOTH windows' shares of d_work
        t_ov = d_work + 2 * wfl; t_hi = t_ov + round16f(nsc * NYQ_HALF_OV); t_de = t_hi + round16f(nsc * kPostHist);
        t_pf[0] = t_de + round16f(nsc); t_pf[1] = t_pf[0] + round16f(nstreams * 6);
        ov = d_overlap ? d_overlap : t_ov; hi = d_hist ? d_hist : t_hi; de = d_deemph ? d_deemph : t_de;
        if (!d_overlap) NYQ_HIP(ctx, hipMemsetAsync(t_ov, 0, nsc * NYQ_HALF_OV * sizeof(float), ctx->stream));
        if (!d_hist) NYQ_HIP(ctx, hipMemsetAsync(t_hi, 0, nsc * kPostHist * sizeof(float), ctx->stream));
        if (!d_deemph) NYQ_HIP(ctx, hipMemsetAsync(t_de, 0, nsc * sizeof(float), ctx->stream));
    }
    hipStream_t main_stream = ctx->stream;
    size_t k = 0;
    for (size_t f0 = 0; f0 < nframes; f0 += W, k++) {
        const size_t len = nframes - f0 < W ? nframes - f0 : W;
        const bool lastw = f0 + len == nframes;
        float *pf_out = lastw ? d_pf_state_out : t_pf[k & 1];
        float *wpcm = overlap ? d_pcm + (k & 1) * pfl : d_pcm, *wwork = overlap ? d_work + (k & 1) * wfl : d_work;
        if (overlap && k >= 2) NYQ_HIP(ctx, hipStreamWaitEvent(main_stream, ctx->ev_pool[2 * (k - 2) + 1], 0));   // post(k-2) has read this buffer
        int rc = synth_core(ctx, LM, d_freq + f0 * channels * N, d_transient ? d_transient + f0 : nullptr, wpcm, ov, ov, wwork,
                            nstreams, len, channels, nframes);
        if (rc != NYQ_OK) return rc;
        if (overlap) {
            NYQ_HIP(ctx, hipEventRecord(ctx->ev_pool[2 * k], main_stream));
            NYQ_HIP(ctx, hipStreamWaitEvent(ctx->s_post, ctx->ev_pool[2 * k], 0));
            ctx->stream = ctx->s_post;
        }
        rc = post_core(ctx, LM, wpcm, d_pf_pitch + f0, d_pf_gain + f0, d_pf_tapset + f0, pf_in, pf_out, hi, de, d_out + f0 * N * channels,
                       nstreams, len, channels, nframes);
        ctx->stream = main_stream;
        if (rc != NYQ_OK) return rc;
        if (overlap) NYQ_HIP(ctx, hipEventRecord(ctx->ev_pool[2 * k + 1], ctx->s_post));
        pf_in = pf_out;
    }
    if (overlap) NYQ_HIP(ctx, hipStreamWaitEvent(main_stream, ctx->ev_pool[2 * (k - 1) + 1], 0));
    return NYQ_OK;
}

extern "C" int nyq_celt_chain_dev(nyq_ctx *ctx, int LM, const float *d_freq, const unsigned char *d_transient,
                                  const int *d_pf_pitch, const float *d_pf_gain, const int *d_pf_tapset,
                                  const float *d_pf_state_in, float *d_pf_state_out, float *d_overlap, float *d_hist,
                                  float *d_deemph, float *d_out, float *d_pcm, float *d_work, size_t nstreams,
                                  size_t nframes, int channels) {
    return chain_core(ctx, LM, d_freq, d_transient, d_pf_pitch, d_pf_gain, d_pf_tapset, d_pf_state_in, d_pf_state_out, d_overlap,
                      d_hist, d_deemph, d_out, d_pcm, d_work, nstreams, nframes, channels, 0);
}

// ---- row f3: the samples straight into the FILE's interleaved layout (opus_multistream_decoder.c:305-331) -------------
static_assert(sizeof(nyq_out_desc) == sizeof(OutDesc) && offsetof(nyq_out_desc, base) == offsetof(OutDesc, base) &&
                  offsetof(nyq_out_desc, first) == offsetof(OutDesc, first) && offsetof(nyq_out_desc, last) == offsetof(OutDesc, last) &&
                  offsetof(nyq_out_desc, t0) == offsetof(OutDesc, t0) && offsetof(nyq_out_desc, cstride) == offsetof(OutDesc, cstride) &&
                  offsetof(nyq_out_desc, coff) == offsetof(OutDesc, coff0) && offsetof(nyq_out_desc, gain) == offsetof(OutDesc, gain),
              "nyq_out_desc (C ABI) and nyq::OutDesc (kernels) are one layout");

extern "C" int nyq_celt_chain_mapped_dev(nyq_ctx *ctx, int LM, const float *d_freq, const unsigned char *d_transient,
                                         const int *d_pf_pitch, const float *d_pf_gain, const int *d_pf_tapset,
                                         const float *d_pf_state_in, float *d_pf_state_out, float *d_overlap, float *d_hist,
                                         float *d_deemph, float *d_out, const nyq_out_desc *d_desc, float *d_pcm, float *d_work,
                                         size_t nstreams, size_t nframes, int channels) {
    if (ctx && !d_out && !d_desc) return fail(ctx, NYQ_ERR_INVALID, "nyq_celt_chain_mapped_dev: neither d_out nor d_desc");
    // (d_out may be null when every stream has a destination: the kernels never touch it then; chain_core wants a non-null
    // pointer for its argument check only)
    float *out = d_out ? d_out : reinterpret_cast<float *>(const_cast<nyq_out_desc *>(d_desc));
    return chain_core(ctx, LM, d_freq, d_transient, d_pf_pitch, d_pf_gain, d_pf_tapset, d_pf_state_in, d_pf_state_out, d_overlap,
                      d_hist, d_deemph, out, d_pcm, d_work, nstreams, nframes, channels, 0, reinterpret_cast<const OutDesc *>(d_desc));
}

// ---- band shapes from symbol records -------------------------------------------------------------------------------
static_assert(sizeof(nyq_sym_head) == sizeof(SymHead) && sizeof(nyq_sym_leaf) == sizeof(SymLeaf) && sizeof(nyq_sym_vec) == sizeof(SymVec) &&
                  sizeof(nyq_sym_op) == sizeof(SymOp) && NYQ_SYM_MAX_OPS == kSymMaxOps && NYQ_SYM_MAX_VECS == kSymMaxVecs &&
                  offsetof(nyq_sym_head, nops) == offsetof(SymHead, nops) && offsetof(nyq_sym_head, lm) == offsetof(SymHead, lm) &&
                  offsetof(nyq_sym_leaf, fold_off) == offsetof(SymLeaf, fold_off) && offsetof(nyq_sym_leaf, img) == offsetof(SymLeaf, img) &&
                  offsetof(nyq_sym_leaf, index) == offsetof(SymLeaf, index) && offsetof(nyq_sym_vec, b_in) == offsetof(SymVec, b_in) &&
                  offsetof(nyq_sym_vec, fill_hi) == offsetof(SymVec, fill_hi) &&
                  offsetof(nyq_sym_op, f1) == offsetof(SymOp, f1),
              "nyq_sym_* (C ABI) and nyq::Sym* (kernel) are one layout");

extern "C" size_t nyq_celt_symbol_bytes(int channels) { return channels == 1 || channels == 2 ? sym_bytes(channels) : 0; }
extern "C" size_t nyq_celt_symbol_bytes_lm(int channels, int LM) {
    return (channels == 1 || channels == 2) && LM >= 0 && LM <= 3 ? sym_bytes(channels, LM) : 0;
}

// sstride / fstride: frames per stream in d_sym / d_freq (0 = nframes: dense)
// d_off: records packed back to back inside each stream's region of sstride slots, d_off[stream * ostride + frame] = byte offset
// of the frame's record from the region's start; null: one record per slot of nyq_celt_symbol_bytes
static int shape_core(nyq_ctx *ctx, const void *d_sym, float *d_freq, size_t nstreams, size_t nframes, int channels, size_t sstride,
                      size_t fstride = 0, const unsigned *d_off = nullptr, size_t ostride = 0, int LM = 3, size_t slot = 0) {
    if (!ctx) return fail(nullptr, NYQ_ERR_INVALID, "nyq_celt_shape_dev: ctx is NULL");
    if (channels != 1 && channels != 2) return fail(ctx, NYQ_ERR_INVALID, "nyq_celt_shape_dev: symbol records carry mono and stereo streams");
    if (LM < 0 || LM > 3) return fail(ctx, NYQ_ERR_INVALID, "nyq_celt_shape_dev: LM must be 0..3");
    if (slot == 0) slot = sym_bytes(channels, LM);
    if (slot < 256 || slot > 65520 || slot % 16 != 0) return fail(ctx, NYQ_ERR_INVALID, "nyq_celt_shape_dev: a record slot is 256..65520 bytes, a multiple of 16");
    if (nstreams == 0 || nframes == 0) return NYQ_OK;
    if (!d_sym || !d_freq) return fail(ctx, NYQ_ERR_INVALID, "nyq_celt_shape_dev: NULL buffer");
    if (sstride == 0) sstride = nframes;
    if (fstride == 0) fstride = nframes;
    if (sstride < nframes || fstride < nframes) return fail(ctx, NYQ_ERR_INVALID, "nyq_celt_shape_dev: a stride is smaller than nframes");
    const size_t total = nstreams * nframes;
    if (total > (size_t)0x7fffffff) return fail(ctx, NYQ_ERR_INVALID, "nyq_celt_shape_dev: more than 2^31 frames in one call");
    NYQ_HIP(ctx, hipSetDevice(ctx->device));
    if (!ctx->d_pvq) {
        std::vector<unsigned> table((size_t)kPvqInfo + kPvqWords);
        if (pvq_table_build(table.data()) != kPvqWords) return fail(ctx, NYQ_ERR_INVALID, "nyq_celt_shape_dev: pulse-vector table has an unexpected size");
        table.push_back(0);                                          // (behind the table: the launch's frame counter)
        NYQ_HIP(ctx, hipMalloc(&ctx->d_pvq, table.size() * sizeof(unsigned)));
        NYQ_HIP(ctx, hipMemcpy(ctx->d_pvq, table.data(), table.size() * sizeof(unsigned), hipMemcpyHostToDevice));
    }
    // a resident grid (one workgroup of kShapeWaves frames per CU: 141 KB of LDS), grid-stride over the frames: the table
    // is staged into LDS once per workgroup
    const size_t want = (total + kShapeWaves - 1) / kShapeWaves, resident = (size_t)ctx->cus;
    const unsigned grid = (unsigned)(want < resident ? want : resident);
    unsigned *d_next = ctx->d_pvq + kPvqInfo + kPvqWords;            // launches of one context are ordered on its stream
    NYQ_HIP(ctx, hipMemsetAsync(d_next, 0, sizeof(unsigned), ctx->stream));
    auto launch = [&](auto kernel) {
        hipLaunchKernelGGL(kernel, dim3(grid), dim3(kWave * kShapeWaves), 0, ctx->stream, static_cast<const unsigned char *>(d_sym), d_freq,
                           ctx->d_pvq, (long)nstreams, (long)nframes, channels, (long)sstride, (long)fstride, d_off, (long)(ostride ? ostride : nframes),
                           d_next, (long)slot);
    };
    switch (LM) {
    case 0: launch(celt_shape_kernel<0>); break;
    case 1: launch(celt_shape_kernel<1>); break;
    case 2: launch(celt_shape_kernel<2>); break;
    default: launch(celt_shape_kernel<3>); break;
    }
    NYQ_HIP(ctx, hipGetLastError());
    return NYQ_OK;
}

extern "C" int nyq_celt_shape_dev(nyq_ctx *ctx, const void *d_sym, float *d_freq, size_t nstreams, size_t nframes, int channels,
                                  size_t sstride) {
    return shape_core(ctx, d_sym, d_freq, nstreams, nframes, channels, sstride);
}
extern "C" int nyq_celt_shape_lm_dev(nyq_ctx *ctx, int LM, const void *d_sym, float *d_freq, size_t nstreams, size_t nframes, int channels,
                                     size_t sstride) {
    return shape_core(ctx, d_sym, d_freq, nstreams, nframes, channels, sstride, 0, nullptr, 0, LM);
}
extern "C" int nyq_celt_shape_slots_dev(nyq_ctx *ctx, int LM, const void *d_sym, size_t slot_bytes, float *d_freq, size_t nstreams, size_t nframes,
                                        int channels) {
    return shape_core(ctx, d_sym, d_freq, nstreams, nframes, channels, 0, 0, nullptr, 0, LM, slot_bytes);
}

// ---- the entropy stage on the device (nyq_entropy_kernel.hpp): frames' bytes -> symbol records ------------------------------------
static_assert(sizeof(nyq_ent_desc) == sizeof(nyq_ent::EntDesc) && sizeof(nyq_ent_info) == sizeof(nyq_ent::EntInfo) &&
                  sizeof(nyq_ent_state) == sizeof(EnergyState) && NYQ_ENT_ENERGY_BYTES == sizeof(nyq_ent::EntEnergy),
              "include/nyq_imdct.h and the kernels agree on the entropy stage's records");
extern "C" size_t nyq_celt_entropy_tables_bytes(void) { return sizeof(nyq_ent::EntropyTables); }
extern "C" size_t nyq_celt_entropy_slot_bytes(int channels, int LM) {
    return (channels == 1 || channels == 2) && LM >= 0 && LM <= 3 ? (size_t)nyq_ent::recFullSlot(channels, LM) : 0;
}
constexpr size_t kByteSlot = 1280;                                   // a frame's bytes in the host-buffer form (a CELT frame is at most 1275)
// frames' bytes -> spread records + infos (+ the per-frame arrays of the chain), all on the context stream
static int entropy_core(nyq_ctx *ctx, int LM, const void *d_tables, const unsigned char *d_payload, size_t payload_bytes, const void *d_desc, size_t pslot,
                        size_t nstreams, size_t nframes, unsigned char *d_sym, size_t slot, nyq_ent::EntInfo *d_info, nyq_ent::EntEnergy *d_energy,
                        EnergyState *d_state, int fresh) {
    const size_t total = nstreams * nframes;
    hipLaunchKernelGGL(celt_entropy_kernel, dim3((unsigned)((total + 63) / 64)), dim3(64), 0, ctx->stream,
                       static_cast<const nyq_ent::EntropyTables *>(d_tables), d_payload, (long)payload_bytes, static_cast<const nyq_ent::EntDesc *>(d_desc), (long)total, LM,
                       d_sym, (long)slot, d_info, d_energy, (long)pslot);
    NYQ_HIP(ctx, hipGetLastError());
    hipLaunchKernelGGL(celt_energy_kernel, dim3((unsigned)nstreams), dim3(64), 0, ctx->stream, static_cast<const nyq_ent::EntropyTables *>(d_tables),
                       d_info, d_energy, d_sym, (long)slot, (long)nstreams, (long)nframes, d_state, fresh ? 1 : 0);
    NYQ_HIP(ctx, hipGetLastError());
    return NYQ_OK;
}
extern "C" int nyq_celt_entropy_dev(nyq_ctx *ctx, int LM, const void *d_tables, const unsigned char *d_payload, size_t payload_bytes, const nyq_ent_desc *d_desc,
                                    size_t nstreams, size_t nframes, int channels, void *d_sym, size_t slot_bytes, nyq_ent_info *d_info, void *d_energy,
                                    nyq_ent_state *d_state, int fresh) {
    if (!ctx) return fail(nullptr, NYQ_ERR_INVALID, "nyq_celt_entropy_dev: ctx is NULL");
    if (channels != 1 && channels != 2) return fail(ctx, NYQ_ERR_INVALID, "nyq_celt_entropy_dev: symbol records carry mono and stereo streams");
    if (LM < 0 || LM > 3) return fail(ctx, NYQ_ERR_INVALID, "nyq_celt_entropy_dev: LM must be 0..3");
    if (nstreams == 0 || nframes == 0) return NYQ_OK;
    if (!d_tables || !d_payload || !d_desc || !d_sym || !d_info || !d_energy || !d_state)
        return fail(ctx, NYQ_ERR_INVALID, "nyq_celt_entropy_dev: NULL buffer");
    if (nstreams * nframes > (size_t)0x7fffffff) return fail(ctx, NYQ_ERR_INVALID, "nyq_celt_entropy_dev: more than 2^31 frames in one call");
    NYQ_HIP(ctx, hipSetDevice(ctx->device));
    const size_t slot = slot_bytes ? slot_bytes : sym_bytes(channels, LM);
    if (slot < 256 || slot > 65520 || slot % 16 != 0) return fail(ctx, NYQ_ERR_INVALID, "nyq_celt_entropy_dev: a record slot is 256..65520 bytes, a multiple of 16");
    return entropy_core(ctx, LM, d_tables, d_payload, payload_bytes, d_desc, 0, nstreams, nframes, static_cast<unsigned char *>(d_sym), slot,
                        reinterpret_cast<nyq_ent::EntInfo *>(d_info), static_cast<nyq_ent::EntEnergy *>(d_energy), reinterpret_cast<EnergyState *>(d_state), fresh);
}
// the tables of the entropy stage, kept by the context for the host-buffer form below (host_tables: nyqh_entropy_tables' block)
extern "C" int nyq_ctx_set_entropy_tables(nyq_ctx *ctx, const void *host_tables, size_t bytes) {
    if (!ctx) return fail(nullptr, NYQ_ERR_INVALID, "nyq_ctx_set_entropy_tables: ctx is NULL");
    if (!host_tables || bytes != sizeof(nyq_ent::EntropyTables)) return fail(ctx, NYQ_ERR_INVALID, "nyq_ctx_set_entropy_tables: the block is nyq_celt_entropy_tables_bytes() long");
    NYQ_HIP(ctx, hipSetDevice(ctx->device));
    if (!ctx->d_ent_tables) NYQ_HIP(ctx, hipMalloc(&ctx->d_ent_tables, bytes));
    NYQ_HIP(ctx, hipMemcpy(ctx->d_ent_tables, host_tables, bytes, hipMemcpyHostToDevice));
    return NYQ_OK;
}

extern "C" int nyq_celt_entropy_split_dev(nyq_ctx *ctx, const nyq_ent_info *d_info, size_t n, unsigned char *d_transient, int *d_pf_pitch,
                                          float *d_pf_gain, int *d_pf_tapset) {
    if (!ctx) return fail(nullptr, NYQ_ERR_INVALID, "nyq_celt_entropy_split_dev: ctx is NULL");
    if (n == 0) return NYQ_OK;
    if (!d_info || !d_transient || !d_pf_pitch || !d_pf_gain || !d_pf_tapset) return fail(ctx, NYQ_ERR_INVALID, "nyq_celt_entropy_split_dev: NULL buffer");
    NYQ_HIP(ctx, hipSetDevice(ctx->device));
    hipLaunchKernelGGL(celt_entropy_split_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, ctx->stream,
                       reinterpret_cast<const nyq_ent::EntInfo *>(d_info), (long)n, d_transient, d_pf_pitch, d_pf_gain, d_pf_tapset);
    NYQ_HIP(ctx, hipGetLastError());
    return NYQ_OK;
}

extern "C" void *nyq_device_alloc(nyq_ctx *ctx, size_t bytes) {
    void *p = nullptr;
    if (!ctx || bytes == 0 || hipSetDevice(ctx->device) != hipSuccess || hipMalloc(&p, bytes) != hipSuccess) return nullptr;
    return p;
}
extern "C" void nyq_device_free(nyq_ctx *ctx, void *p) {
    if (ctx && p) {
        (void)hipSetDevice(ctx->device);
        (void)hipFree(p);
    }
}
extern "C" int nyq_device_zero(nyq_ctx *ctx, void *d_p, size_t bytes) {
    if (!ctx || !d_p) return fail(ctx, NYQ_ERR_INVALID, "nyq_device_zero: NULL argument");
    NYQ_HIP(ctx, hipSetDevice(ctx->device));
    NYQ_HIP(ctx, hipMemsetAsync(d_p, 0, bytes, ctx->stream));
    NYQ_HIP(ctx, hipStreamSynchronize(ctx->stream));
    return NYQ_OK;
}
extern "C" int nyq_device_download(nyq_ctx *ctx, void *host_dst, const void *d_src, size_t bytes) {
    if (!ctx || !host_dst || !d_src) return fail(ctx, NYQ_ERR_INVALID, "nyq_device_download: NULL argument");
    NYQ_HIP(ctx, hipSetDevice(ctx->device));
    NYQ_HIP(ctx, hipMemcpyAsync(host_dst, d_src, bytes, hipMemcpyDeviceToHost, ctx->stream));
    NYQ_HIP(ctx, hipStreamSynchronize(ctx->stream));
    return NYQ_OK;
}
// one destination channel repeated in another (a channel mapping may name one decoded channel twice)
__global__ void dup_channel_kernel(float *base, int cstride, int src, int dst, long n) {
    for (long t = (long)blockIdx.x * blockDim.x + threadIdx.x; t < n; t += (long)gridDim.x * blockDim.x) base[t * cstride + dst] = base[t * cstride + src];
}
extern "C" int nyq_device_dup_channel(nyq_ctx *ctx, float *d_base, int cstride, int src_slot, int dst_slot, size_t nsamples) {
    if (!ctx || !d_base || cstride < 1 || src_slot < 0 || dst_slot < 0 || src_slot >= cstride || dst_slot >= cstride)
        return fail(ctx, NYQ_ERR_INVALID, "nyq_device_dup_channel: bad argument");
    if (nsamples == 0 || src_slot == dst_slot) return NYQ_OK;
    NYQ_HIP(ctx, hipSetDevice(ctx->device));
    hipLaunchKernelGGL(dup_channel_kernel, dim3((unsigned)std::min<size_t>((nsamples + 255) / 256, 4096)), dim3(256), 0, ctx->stream, d_base,
                       cstride, src_slot, dst_slot, (long)nsamples);
    NYQ_HIP(ctx, hipGetLastError());
    NYQ_HIP(ctx, hipStreamSynchronize(ctx->stream));
    return NYQ_OK;
}

// ---- Vorbis inverse MDCT ---------------------------------------------------------------------
constexpr int kVorbisWavesPerBlock = 2;
static int need_scratch(nyq_ctx *ctx, size_t bytes);
static size_t round16f(size_t nfloats);

template <int LOGN4>
static int launch_vorbis(nyq_ctx *ctx, const float *d_in, float *d_out, size_t batch) {
    using V = VGeo<LOGN4>;
    VTables T{ctx->d_vtab + ctx->vrot_off[LOGN4], ctx->d_vtab + ctx->vtw_off[LOGN4]};
    if (ctx->res_vorbis[LOGN4] == 0) {
        int per_cu = 0;
        hipError_t e = hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, vorbis_imdct_kernel<LOGN4, kVorbisWavesPerBlock>,
                                                                    kWave * kVorbisWavesPerBlock, 0);
        if (e != hipSuccess || per_cu < 1) per_cu = 1;
        if (per_cu > 4) per_cu = 4;
        ctx->res_vorbis[LOGN4] = per_cu * ctx->cus;
    }
    const size_t ngroups = (batch + V::G - 1) / V::G;
    const size_t need = (ngroups + kVorbisWavesPerBlock - 1) / kVorbisWavesPerBlock;
    const unsigned grid = (unsigned)(need < (size_t)ctx->res_vorbis[LOGN4] ? need : (size_t)ctx->res_vorbis[LOGN4]);
    hipLaunchKernelGGL((vorbis_imdct_kernel<LOGN4, kVorbisWavesPerBlock>), dim3(grid), dim3(kWave * kVorbisWavesPerBlock), 0,
                       ctx->stream, d_in, d_out, (long)batch, T);
    NYQ_HIP(ctx, hipGetLastError());
    return NYQ_OK;
}

extern "C" int nyq_vorbis_imdct_batch_dev(nyq_ctx *ctx, int n, const float *d_in, float *d_out, size_t batch) {
    if (!ctx) return fail(nullptr, NYQ_ERR_INVALID, "nyq_vorbis_imdct_batch_dev: ctx is NULL");
    int m = -1;
    for (int k = 4; k <= 11; k++)
        if (n == (4 << k)) m = k;
    if (m < 0) return fail(ctx, NYQ_ERR_INVALID, "nyq_vorbis_imdct_batch_dev: n must be a power of two in 64..8192");
    if (batch == 0) return NYQ_OK;
    if (!d_in || !d_out || d_in == d_out) return fail(ctx, NYQ_ERR_INVALID, "nyq_vorbis_imdct_batch_dev: NULL or aliased buffers");
    if (!aligned16(d_in) || !aligned16(d_out))
        return fail(ctx, NYQ_ERR_INVALID, "nyq_vorbis_imdct_batch_dev: device pointers must be 16-byte aligned");
    switch (m) {
    case 4: return launch_vorbis<4>(ctx, d_in, d_out, batch);
    case 5: return launch_vorbis<5>(ctx, d_in, d_out, batch);
    case 6: return launch_vorbis<6>(ctx, d_in, d_out, batch);
    case 7: return launch_vorbis<7>(ctx, d_in, d_out, batch);
    case 8: return launch_vorbis<8>(ctx, d_in, d_out, batch);
    case 9: return launch_vorbis<9>(ctx, d_in, d_out, batch);
    case 10: return launch_vorbis<10>(ctx, d_in, d_out, batch);
    default: return launch_vorbis<11>(ctx, d_in, d_out, batch);
    }
}

extern "C" int nyq_vorbis_imdct_batch(nyq_ctx *ctx, int n, const float *in, float *out, size_t batch) {
    if (!ctx) return fail(nullptr, NYQ_ERR_INVALID, "nyq_vorbis_imdct_batch: ctx is NULL");
    if (batch == 0) return NYQ_OK;
    if (!in || !out || n < 64 || n > 8192) return fail(ctx, NYQ_ERR_INVALID, "nyq_vorbis_imdct_batch: bad argument");
    NYQ_HIP(ctx, hipSetDevice(ctx->device));
    const size_t n_in = round16f(batch * (size_t)(n / 2)), n_out = round16f(batch * (size_t)n);
    int rc = need_scratch(ctx, (n_in + n_out) * sizeof(float));
    if (rc != NYQ_OK) return rc;
    float *d_in = ctx->d_scratch, *d_out = d_in + n_in;
    NYQ_HIP(ctx, hipMemcpyAsync(d_in, in, batch * (size_t)(n / 2) * sizeof(float), hipMemcpyHostToDevice, ctx->stream));
    rc = nyq_vorbis_imdct_batch_dev(ctx, n, d_in, d_out, batch);
    if (rc != NYQ_OK) return rc;
    NYQ_HIP(ctx, hipMemcpyAsync(out, d_out, batch * (size_t)n * sizeof(float), hipMemcpyDeviceToHost, ctx->stream));
    NYQ_HIP(ctx, hipStreamSynchronize(ctx->stream));
    return NYQ_OK;
}

// ---- host-buffer variants ---------------------------------------------------------
static int need_scratch(nyq_ctx *ctx, size_t bytes) {
    if (bytes <= ctx->scratch_bytes) return NYQ_OK;
    if (ctx->d_scratch) {
        NYQ_HIP(ctx, hipStreamSynchronize(ctx->stream));
        NYQ_HIP(ctx, hipFree(ctx->d_scratch));
        ctx->d_scratch = nullptr;
        ctx->scratch_bytes = 0;
    }
    size_t want = bytes + bytes / 4 + 4096;
    hipError_t e = hipMalloc(&ctx->d_scratch, want);
    if (e != hipSuccess) return fail(ctx, NYQ_ERR_ALLOC, std::string("scratch hipMalloc: ") + hipGetErrorString(e));
    ctx->scratch_bytes = want;
    return NYQ_OK;
}

static size_t round16f(size_t nfloats) { return (nfloats + 3) & ~(size_t)3; }

extern "C" int nyq_ifft_batch(nyq_ctx *ctx, int nfft, const float *in, float *out, size_t batch) {
    if (!ctx) return fail(nullptr, NYQ_ERR_INVALID, "nyq_ifft_batch: ctx is NULL");
    if (nfft != 480 && nfft != 240 && nfft != 120 && nfft != 60)
        return fail(ctx, NYQ_ERR_INVALID, "nyq_ifft_batch: nfft must be 480, 240, 120 or 60");
    if (batch == 0) return NYQ_OK;
    if (!in || !out) return fail(ctx, NYQ_ERR_INVALID, "nyq_ifft_batch: NULL in/out");
    NYQ_HIP(ctx, hipSetDevice(ctx->device));
    const size_t n = round16f(batch * 2 * (size_t)nfft);
    int rc = need_scratch(ctx, 2 * n * sizeof(float));
    if (rc != NYQ_OK) return rc;
    float *d_in = ctx->d_scratch, *d_out = ctx->d_scratch + n;
    NYQ_HIP(ctx, hipMemcpyAsync(d_in, in, batch * 2 * nfft * sizeof(float), hipMemcpyHostToDevice, ctx->stream));
    rc = nyq_ifft_batch_dev(ctx, nfft, d_in, d_out, batch);
    if (rc != NYQ_OK) return rc;
    NYQ_HIP(ctx, hipMemcpyAsync(out, d_out, batch * 2 * nfft * sizeof(float), hipMemcpyDeviceToHost, ctx->stream));
    NYQ_HIP(ctx, hipStreamSynchronize(ctx->stream));
    return NYQ_OK;
}

// Pinned host memory for callers of the host-buffer entry points: copies from/to such buffers are DMA
// transfers that overlap the kernels; pageable buffers work too but are staged by the runtime.
extern "C" void *nyq_host_alloc(size_t bytes) {
    void *p = nullptr;
    // portable: every device of the process may DMA from / to it (one decoder can feed several GPUs)
    if (bytes == 0 || hipHostMalloc(&p, bytes, hipHostMallocPortable) != hipSuccess) return nullptr;
    return p;
}

extern "C" void nyq_host_free(void *p) {
    if (p) (void)hipHostFree(p);
}

// Wait for a stream without burning a host core: callers of the host-pointer entry points run next to CPU
// threads that have real work (the entropy stage), and hipStreamSynchronize spins.
static int wait_blocking(nyq_ctx *ctx, hipStream_t st) {
    if (!ctx->ev_block) NYQ_HIP(ctx, hipEventCreateWithFlags(&ctx->ev_block, hipEventBlockingSync | hipEventDisableTiming));
    NYQ_HIP(ctx, hipEventRecord(ctx->ev_block, st));
    NYQ_HIP(ctx, hipEventSynchronize(ctx->ev_block));
    return NYQ_OK;
}

static int need_copy_streams(nyq_ctx *ctx, size_t nevents) {
    if (!ctx->s_h2d) NYQ_HIP(ctx, hipStreamCreateWithFlags(&ctx->s_h2d, hipStreamNonBlocking));
    if (!ctx->s_d2h) NYQ_HIP(ctx, hipStreamCreateWithFlags(&ctx->s_d2h, hipStreamNonBlocking));
    while (ctx->ev_pool.size() < nevents) {
        hipEvent_t e;
        NYQ_HIP(ctx, hipEventCreateWithFlags(&e, hipEventDisableTiming));
        ctx->ev_pool.push_back(e);
    }
    return NYQ_OK;
}

// Whatever way a pipelined host-pointer call returns (an error half way included), no copy that still
// reads or writes the CALLER's buffers may be left in flight.
struct DrainOnExit {
    nyq_ctx *c;
    ~DrainOnExit() {
        if (c->s_h2d) (void)hipStreamSynchronize(c->s_h2d);
        (void)hipStreamSynchronize(c->stream);
        if (c->s_d2h) (void)hipStreamSynchronize(c->s_d2h);
    }
};

// Host rows -> device -> host, cut into pieces of whole chains so that the upload of piece k+1, the kernel
// of piece k and the download of piece k-1 run at the same time (three streams, events between them).
constexpr size_t kHostPieceBytes = (size_t)32 << 20;   // of input per piece
constexpr size_t kHostMaxPieces = 64;
constexpr size_t kHostWindowBytes = (size_t)24 << 20;  // of input per time window (frames -> PCM calls on few long streams)
constexpr size_t kHostMaxWindows = 256;

static int imdct_host(nyq_ctx *ctx, int shift, const float *in, const float *carry, size_t carry_rows, float *fin,
                      float *tail, size_t tail_rows, size_t nchains, size_t len, bool chain) {
    const size_t rows = nchains * len;
    const size_t n2 = (size_t)(NYQ_MDCT_N >> (shift + 1));
    NYQ_HIP(ctx, hipSetDevice(ctx->device));
    const size_t n_in = round16f(rows * n2), n_c = round16f(carry_rows * NYQ_HALF_OV),
                 n_t = round16f((rows + nchains) * NYQ_HALF_OV), n_to = round16f(tail_rows * NYQ_HALF_OV);
    int rc = need_scratch(ctx, (2 * n_in + n_c + n_t + n_to) * sizeof(float));
    if (rc != NYQ_OK) return rc;
    float *d_in = ctx->d_scratch, *d_fin = d_in + n_in, *d_c = d_fin + n_in, *d_t = d_c + n_c, *d_to = d_t + n_t;
    // chains per piece: whole chains, a multiple of 4 rows so that device pointers stay 16-byte aligned
    const size_t chain_bytes = len * n2 * sizeof(float);
    size_t per = (kHostPieceBytes + chain_bytes - 1) / chain_bytes;
    if ((nchains + per - 1) / per > kHostMaxPieces) per = (nchains + kHostMaxPieces - 1) / kHostMaxPieces;
    per = (per + 3) & ~(size_t)3;
    const size_t npieces = (nchains + per - 1) / per;
    rc = need_copy_streams(ctx, 2 * npieces + 1);
    if (rc != NYQ_OK) return rc;
    DrainOnExit drain{ctx};
    // the copy streams start after whatever the caller queued on the compute stream before this call
    hipEvent_t ev0 = ctx->ev_pool[2 * npieces];
    NYQ_HIP(ctx, hipEventRecord(ev0, ctx->stream));
    NYQ_HIP(ctx, hipStreamWaitEvent(ctx->s_h2d, ev0, 0));
    NYQ_HIP(ctx, hipStreamWaitEvent(ctx->s_d2h, ev0, 0));
    const size_t work_per_chain = (len + 1) * NYQ_HALF_OV;   // nyq_imdct_chain_dev's d_work layout
    for (size_t k = 0; k < npieces; k++) {
        const size_t c0 = k * per, nc = (nchains - c0 < per ? nchains - c0 : per), r0 = c0 * len, nr = nc * len;
        hipEvent_t up = ctx->ev_pool[2 * k], done = ctx->ev_pool[2 * k + 1];
        NYQ_HIP(ctx, hipMemcpyAsync(d_in + r0 * n2, in + r0 * n2, nr * n2 * sizeof(float), hipMemcpyHostToDevice, ctx->s_h2d));
        if (carry)
            NYQ_HIP(ctx, hipMemcpyAsync(d_c + c0 * NYQ_HALF_OV, carry + c0 * NYQ_HALF_OV, nc * NYQ_HALF_OV * sizeof(float),
                                        hipMemcpyHostToDevice, ctx->s_h2d));
        NYQ_HIP(ctx, hipEventRecord(up, ctx->s_h2d));
        NYQ_HIP(ctx, hipStreamWaitEvent(ctx->stream, up, 0));
        if (chain)
            rc = nyq_imdct_chain_dev(ctx, shift, d_in + r0 * n2, carry ? d_c + c0 * NYQ_HALF_OV : nullptr, d_fin + r0 * n2,
                                     tail ? d_to + c0 * NYQ_HALF_OV : nullptr, d_t + c0 * work_per_chain, nc, len);
        else
            rc = nyq_imdct_batch_dev(ctx, shift, d_in + r0 * n2, carry ? d_c + c0 * NYQ_HALF_OV : nullptr, d_fin + r0 * n2,
                                     tail ? d_t + c0 * NYQ_HALF_OV : nullptr, nr);
        if (rc != NYQ_OK) {
            (void)hipStreamSynchronize(ctx->s_h2d);
            (void)hipStreamSynchronize(ctx->stream);
            (void)hipStreamSynchronize(ctx->s_d2h);
            return rc;
        }
        NYQ_HIP(ctx, hipEventRecord(done, ctx->stream));
        NYQ_HIP(ctx, hipStreamWaitEvent(ctx->s_d2h, done, 0));
        NYQ_HIP(ctx, hipMemcpyAsync(fin + r0 * n2, d_fin + r0 * n2, nr * n2 * sizeof(float), hipMemcpyDeviceToHost, ctx->s_d2h));
        if (tail)
            NYQ_HIP(ctx, hipMemcpyAsync(tail + c0 * NYQ_HALF_OV, (chain ? d_to : d_t) + c0 * NYQ_HALF_OV,
                                        nc * NYQ_HALF_OV * sizeof(float), hipMemcpyDeviceToHost, ctx->s_d2h));
    }
    if ((rc = wait_blocking(ctx, ctx->s_d2h)) != NYQ_OK) return rc;
    if ((rc = wait_blocking(ctx, ctx->stream)) != NYQ_OK) return rc;
    return NYQ_OK;
}

extern "C" int nyq_imdct_batch(nyq_ctx *ctx, int shift, const float *in, const float *carry, float *fin, float *tail,
                               size_t batch) {
    if (!ctx) return fail(nullptr, NYQ_ERR_INVALID, "nyq_imdct_batch: ctx is NULL");
    if (shift < 0 || shift > 3) return fail(ctx, NYQ_ERR_INVALID, "nyq_imdct_batch: shift must be 0..3");
    if (batch == 0) return NYQ_OK;
    if (!in || !fin) return fail(ctx, NYQ_ERR_INVALID, "nyq_imdct_batch: NULL in/fin");
    return imdct_host(ctx, shift, in, carry, batch, fin, tail, batch, batch, 1, false);
}

extern "C" int nyq_imdct_chain(nyq_ctx *ctx, int shift, const float *in, const float *carry0, float *pcm,
                               float *tail_out, size_t nchains, size_t len) {
    if (!ctx) return fail(nullptr, NYQ_ERR_INVALID, "nyq_imdct_chain: ctx is NULL");
    if (shift < 0 || shift > 3) return fail(ctx, NYQ_ERR_INVALID, "nyq_imdct_chain: shift must be 0..3");
    if (nchains * len == 0) return NYQ_OK;
    if (!in || !pcm) return fail(ctx, NYQ_ERR_INVALID, "nyq_imdct_chain: NULL in/pcm");
    return imdct_host(ctx, shift, in, carry0, nchains, pcm, tail_out, nchains, nchains, len, true);
}

extern "C" int nyq_celt_synth(nyq_ctx *ctx, int LM, const float *freq, const unsigned char *transient, float *pcm,
                              float *state, size_t nstreams, size_t nframes, int channels) {
    if (!ctx) return fail(nullptr, NYQ_ERR_INVALID, "nyq_celt_synth: ctx is NULL");
    if (LM < 0 || LM > 3) return fail(ctx, NYQ_ERR_INVALID, "nyq_celt_synth: LM must be 0..3");
    if (channels < 1 || channels > 255) return fail(ctx, NYQ_ERR_INVALID, "nyq_celt_synth: channels must be 1..255");
    if (nstreams == 0 || nframes == 0) return NYQ_OK;
    if (!freq || !pcm) return fail(ctx, NYQ_ERR_INVALID, "nyq_celt_synth: NULL freq/pcm");
    NYQ_HIP(ctx, hipSetDevice(ctx->device));
    const size_t N = (size_t)120 << LM, nsc = nstreams * (size_t)channels;
    const size_t n_x = round16f(nsc * nframes * N), n_w = round16f(nyq_celt_synth_work_floats(nstreams, nframes, channels)),
                 n_s = round16f(nsc * NYQ_HALF_OV), n_t = round16f((nstreams * nframes + 3) / 4);
    int rc = need_scratch(ctx, (2 * n_x + n_w + n_s + n_t) * sizeof(float));
    if (rc != NYQ_OK) return rc;
    float *d_x = ctx->d_scratch, *d_p = d_x + n_x, *d_w = d_p + n_x, *d_s = d_w + n_w;
    unsigned char *d_t = reinterpret_cast<unsigned char *>(d_s + n_s);
    NYQ_HIP(ctx, hipMemcpyAsync(d_x, freq, nsc * nframes * N * sizeof(float), hipMemcpyHostToDevice, ctx->stream));
    if (transient) NYQ_HIP(ctx, hipMemcpyAsync(d_t, transient, nstreams * nframes, hipMemcpyHostToDevice, ctx->stream));
    if (state) NYQ_HIP(ctx, hipMemcpyAsync(d_s, state, nsc * NYQ_HALF_OV * sizeof(float), hipMemcpyHostToDevice, ctx->stream));
    rc = nyq_celt_synth_dev(ctx, LM, d_x, transient ? d_t : nullptr, d_p, state ? d_s : nullptr, d_w, nstreams, nframes, channels);
    if (rc != NYQ_OK) return rc;
    NYQ_HIP(ctx, hipMemcpyAsync(pcm, d_p, nsc * nframes * N * sizeof(float), hipMemcpyDeviceToHost, ctx->stream));
    if (state) NYQ_HIP(ctx, hipMemcpyAsync(state, d_s, nsc * NYQ_HALF_OV * sizeof(float), hipMemcpyDeviceToHost, ctx->stream));
    NYQ_HIP(ctx, hipStreamSynchronize(ctx->stream));
    return NYQ_OK;
}

extern "C" size_t nyq_celt_state_floats(size_t nstreams, int channels) {
    const size_t nsc = nstreams * (size_t)(channels > 0 ? channels : 0);
    return nsc * (NYQ_HALF_OV + kPostHist + 1) + nstreams * 6 + nstreams * (sizeof(EnergyState) / sizeof(float));   // (+ the entropy stage's state: the bytes form)
}

// Host arrays may be windows into longer per-stream arrays: consecutive streams are `hstride` frames apart
// (hstride == nframes: dense).  The device side is always dense; the strided case moves its data with 2-D copies.
static hipError_t copy_rows(void *dst, size_t dpitch, const void *src, size_t spitch, size_t width, size_t rows,
                            hipMemcpyKind kind, hipStream_t st) {
    if (dpitch == width && spitch == width) return hipMemcpyAsync(dst, src, width * rows, kind, st);
    return hipMemcpy2DAsync(dst, dpitch, src, spitch, width, rows, kind, st);
}

// desc: per-stream destinations (HOST array of nstreams records whose `base` are DEVICE pointers) or null.  A stream with a
// destination is written there by the kernels and is not downloaded; when EVERY stream has one, `out` may be null.
// Packed symbol records, host -> device: the bytes of frames [f0, f0 + len) of every stream are one contiguous range per
// stream, at a different place in each.  One launch moves them all (a workgroup row per stream reads the page-locked host
// memory over PCIe): per-stream hipMemcpyAsync calls -- 16 small copies per time window -- ran one after the other at 18 GB/s
// and kept the windows of a call from overlapping (profiles/r04_ae_*).
__global__ __launch_bounds__(256) void gather_records_kernel(unsigned char *__restrict__ dst, size_t dst_stream_bytes,
                                                             const unsigned char *__restrict__ src, size_t src_stream_bytes,
                                                             const unsigned *__restrict__ rel16, const unsigned *__restrict__ first16,
                                                             long ostride, long f0, long len) {
    const long k = blockIdx.y;
    const size_t lo = (size_t)rel16[k * ostride + f0] * 16, hi = (size_t)rel16[k * ostride + f0 + len] * 16;
    const uint4 *s = reinterpret_cast<const uint4 *>(src + (size_t)k * src_stream_bytes + (size_t)first16[k] * 16 + lo);
    uint4 *d = reinterpret_cast<uint4 *>(dst + (size_t)k * dst_stream_bytes + lo);
    const size_t n = (hi - lo) / 16;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) d[i] = s[i];
}

// sym: the host input is symbol records (nyq_celt_symbol_bytes(channels) per frame) in place of freq[]: uploaded to their own
// scratch region, turned into freq[] by the shape kernel on the context stream, then the same chain.
static int frames_to_pcm_core(nyq_ctx *ctx, int LM, const float *freq, const unsigned char *transient,
                              const int *pf_pitch, const float *pf_gain, const int *pf_tapset, float *out,
                              float *state, size_t nstreams, size_t nframes, int channels, size_t hstride,
                              const nyq_out_desc *desc = nullptr, const unsigned char *sym = nullptr, const unsigned *offsets = nullptr,
                              size_t sym_stream_bytes = 0, const unsigned *fdesc = nullptr) {
    // fdesc != NULL: `sym` holds the frames' BYTES (slots of kByteSlot) and fdesc a word per frame (len | channels << 16 | end band
    // << 24): the entropy stage runs on the device (nyq_entropy_kernel.hpp) and the per-frame arrays come from it
    const bool bytes = fdesc != nullptr;
    if (!ctx) return fail(nullptr, NYQ_ERR_INVALID, "nyq_celt_frames_to_pcm: ctx is NULL");
    if (LM < 0 || LM > 3) return fail(ctx, NYQ_ERR_INVALID, "nyq_celt_frames_to_pcm: LM must be 0..3");
    if (channels < 1 || channels > 255) return fail(ctx, NYQ_ERR_INVALID, "nyq_celt_frames_to_pcm: channels must be 1..255");
    if (nstreams == 0 || nframes == 0) return NYQ_OK;
    bool all_mapped = desc != nullptr;
    for (size_t k = 0; desc && k < nstreams; k++) all_mapped = all_mapped && desc[k].base != nullptr;
    if ((!freq && !sym) || (!bytes && (!pf_pitch || !pf_gain || !pf_tapset)) || (!out && !all_mapped))
        return fail(ctx, NYQ_ERR_INVALID, "nyq_celt_frames_to_pcm: NULL buffer");
    if (bytes && (!sym || offsets)) return fail(ctx, NYQ_ERR_INVALID, "nyq_celt_bytes_to_pcm_mapped: NULL buffer");
    if (bytes && !ctx->d_ent_tables) return fail(ctx, NYQ_ERR_INVALID, "nyq_celt_bytes_to_pcm_mapped: nyq_ctx_set_entropy_tables first");
    if (desc && channels > 2) return fail(ctx, NYQ_ERR_INVALID, "nyq_celt_frames_to_pcm_mapped: output descriptors serve mono and stereo streams");
    if (sym && channels > 2) return fail(ctx, NYQ_ERR_INVALID, "nyq_celt_symbols_to_pcm_mapped: symbol records carry mono and stereo streams");
    const size_t rec = bytes ? (size_t)nyq_ent::recFullSlot(channels, LM) : sym ? sym_bytes(channels, LM) : 0;   // a record on the device
    NYQ_HIP(ctx, hipSetDevice(ctx->device));
    const size_t N = (size_t)120 << LM, nsc = nstreams * (size_t)channels, nfr = nstreams * nframes;
    const size_t n_x = round16f(nsc * nframes * N), n_w = round16f(nyq_celt_synth_work_floats(nstreams, nframes, channels)),
                 n_p = round16f(nfr), n_t = round16f((nfr + 3) / 4);
    // state on the device: overlap, hist, deemph, pf in, pf out (each 16-byte aligned)
    const size_t n_ov = round16f(nsc * NYQ_HALF_OV), n_hi = round16f(nsc * kPostHist), n_de = round16f(nsc),
                 n_pf = round16f(nstreams * 6);
    const size_t n_ds = desc ? round16f(nstreams * sizeof(nyq_out_desc) / sizeof(float)) : 0;
    const size_t n_sy = sym ? round16f(nfr * rec / sizeof(float)) : 0;
    const size_t n_dw = desc ? kHostMaxWindows * n_ds : 0;           // (the time-window form: a set of records per window)
    const size_t n_of = offsets ? round16f(nfr + 2 * nstreams) : 0;  // packed records: nframes + 1 device offsets per stream, its first
    // the bytes form: the frames' bytes, their descriptor words, infos, energy deltas, the entropy stage's state
    const size_t n_by = bytes ? round16f(nfr * kByteSlot / sizeof(float)) : 0, n_fd = bytes ? round16f(nfr) : 0, n_in = bytes ? round16f(nfr * 4) : 0,
                 n_en = bytes ? round16f(nfr * sizeof(nyq_ent::EntEnergy) / sizeof(float)) : 0,
                 n_es = bytes ? round16f(nstreams * sizeof(EnergyState) / sizeof(float)) : 0;
    int rc = need_scratch(ctx, (3 * n_x + n_w + 3 * n_p + n_t + n_ov + n_hi + n_de + 2 * n_pf + n_ds + n_sy + n_dw + n_of + n_by + n_fd + n_in + n_en + n_es) * sizeof(float));
    if (rc != NYQ_OK) return rc;
    float *d_x = ctx->d_scratch, *d_pcm = d_x + n_x, *d_out = d_pcm + n_x, *d_w = d_out + n_x, *d_pg = d_w + n_w;
    int *d_pp = reinterpret_cast<int *>(d_pg + n_p), *d_pt = d_pp + n_p;
    unsigned char *d_t = reinterpret_cast<unsigned char *>(d_pt + n_p);
    float *d_ov = reinterpret_cast<float *>(d_pt + n_p) + n_t, *d_hi = d_ov + n_ov, *d_de = d_hi + n_hi,
          *d_pfi = d_de + n_de, *d_pfo = d_pfi + n_pf;
    nyq_out_desc *d_ds = desc ? reinterpret_cast<nyq_out_desc *>(d_pfo + n_pf) : nullptr;
    unsigned char *d_sy = sym ? reinterpret_cast<unsigned char *>(d_pfo + n_pf + n_ds) : nullptr;
    unsigned *d_of = offsets ? reinterpret_cast<unsigned *>(d_pfo + n_pf + n_ds + n_sy + n_dw) : nullptr;
    float *d_ent0 = d_pfo + n_pf + n_ds + n_sy + n_dw + n_of;
    unsigned char *d_by = reinterpret_cast<unsigned char *>(d_ent0);
    unsigned *d_fd = reinterpret_cast<unsigned *>(d_ent0 + n_by);
    nyq_ent::EntInfo *d_in = reinterpret_cast<nyq_ent::EntInfo *>(d_ent0 + n_by + n_fd);
    nyq_ent::EntEnergy *d_en = reinterpret_cast<nyq_ent::EntEnergy *>(d_ent0 + n_by + n_fd + n_in);
    EnergyState *d_es = reinterpret_cast<EnergyState *>(d_ent0 + n_by + n_fd + n_in + n_en);
    constexpr size_t kEsFloats = sizeof(EnergyState) / sizeof(float);
    // the bytes form, for streams [s0, s0 + cnt) (all frames of the call): upload on hs, then -- behind `up` -- entropy stage and split
    auto up_bytes = [&](size_t s0, size_t cnt, hipStream_t hs, float *h_es) -> hipError_t {
        hipError_t e = copy_rows(d_by + s0 * nframes * kByteSlot, nframes * kByteSlot, sym + s0 * hstride * kByteSlot, hstride * kByteSlot,
                                 nframes * kByteSlot, cnt, hipMemcpyHostToDevice, hs);
        if (e != hipSuccess) return e;
        e = copy_rows(d_fd + s0 * nframes, nframes * 4, fdesc + s0 * hstride, hstride * 4, nframes * 4, cnt, hipMemcpyHostToDevice, hs);
        if (e != hipSuccess) return e;
        if (h_es) return hipMemcpyAsync(d_es + s0, h_es + s0 * kEsFloats, cnt * sizeof(EnergyState), hipMemcpyHostToDevice, hs);
        return hipMemsetAsync(d_es + s0, 0, cnt * sizeof(EnergyState), hs);
    };
    auto run_entropy = [&](size_t s0, size_t cnt) -> int {
        int r = entropy_core(ctx, LM, ctx->d_ent_tables, d_by + s0 * nframes * kByteSlot, 0, d_fd + s0 * nframes, kByteSlot, cnt, nframes,
                             d_sy + s0 * nframes * rec, rec, d_in + s0 * nframes, d_en + s0 * nframes, d_es + s0, 0);
        if (r != NYQ_OK) return r;
        hipLaunchKernelGGL(celt_entropy_split_kernel, dim3((unsigned)((cnt * nframes + 255) / 256)), dim3(256), 0, ctx->stream, d_in + s0 * nframes,
                           (long)(cnt * nframes), d_t + s0 * nframes, d_pp + s0 * nframes, d_pg + s0 * nframes, d_pt + s0 * nframes);
        NYQ_HIP(ctx, hipGetLastError());
        return NYQ_OK;
    };
    // Packed records (offsets[stream][hstride + 1], 16-byte units from the stream's base sym + stream * stream_bytes): on the device a
    // stream keeps a region of nframes slots and its records sit packed from the region's start; up_sym uploads frames
    // [f0, f0 + len) of streams [s0, s0 + cnt) -- one copy per stream -- or, for slots, one strided copy
    std::vector<unsigned> h_off;                                     // [nstreams][nframes + 1] relative, then [nstreams] first
    const size_t ostr = nframes + 1;
    unsigned *d_first = d_of ? d_of + nstreams * ostr : nullptr;
    const unsigned char *sym_dev = nullptr;                          // the host records as the device sees them (page-locked memory)
    if (offsets) {
        h_off.resize(nstreams * ostr + nstreams);
        for (size_t k = 0; k < nstreams; k++) {
            const unsigned *o = offsets + k * (hstride + 1);
            for (size_t f = 0; f <= nframes; f++) h_off[k * ostr + f] = o[f] - o[0];
            h_off[nstreams * ostr + k] = o[0];
            if ((size_t)(o[nframes] - o[0]) * 16 > nframes * rec) return fail(ctx, NYQ_ERR_INVALID, "nyq_celt_symbols_to_pcm_mapped: a stream's records exceed nframes slots");
        }
        void *dp = nullptr;
        if (hipHostGetDevicePointer(&dp, const_cast<unsigned char *>(sym), 0) == hipSuccess) sym_dev = static_cast<const unsigned char *>(dp);
        else (void)hipGetLastError();                                // (ordinary memory: per-stream copies below)
    }
    auto up_sym = [&](size_t s0, size_t cnt, size_t f0, size_t len, hipStream_t hs) -> hipError_t {
        if (!offsets)
            return copy_rows(d_sy + (s0 * nframes + f0) * rec, nframes * rec, sym + (s0 * hstride + f0) * rec, hstride * rec, len * rec, cnt,
                             hipMemcpyHostToDevice, hs);
        if (sym_dev) {                                               // (behind the offset tables' upload on the same stream)
            hipLaunchKernelGGL(gather_records_kernel, dim3(24, (unsigned)cnt), dim3(256), 0, hs, d_sy + s0 * nframes * rec, nframes * rec,
                               sym_dev + s0 * sym_stream_bytes, sym_stream_bytes, d_of + s0 * ostr, d_first + s0, (long)ostr, (long)f0, (long)len);
            return hipGetLastError();
        }
        for (size_t k = s0; k < s0 + cnt; k++) {
            const unsigned *o = offsets + k * (hstride + 1);
            const size_t lo = (size_t)o[f0] * 16, hi = (size_t)o[f0 + len] * 16;
            if (hi == lo) continue;
            hipError_t e = hipMemcpyAsync(d_sy + k * nframes * rec + (lo - (size_t)o[0] * 16), sym + k * sym_stream_bytes + lo, hi - lo,
                                          hipMemcpyHostToDevice, hs);
            if (e != hipSuccess) return e;
        }
        return hipSuccess;
    };
    float *h_ov = state, *h_hi = state ? h_ov + nsc * NYQ_HALF_OV : nullptr, *h_de = state ? h_hi + nsc * kPostHist : nullptr,
          *h_pf = state ? h_de + nsc : nullptr, *h_es = state ? h_pf + nstreams * 6 : nullptr;
    // pieces of whole streams: upload of piece k+1, kernels of piece k, download of piece k-1 at the same time
    const size_t stream_bytes = nframes * (size_t)channels * N * sizeof(float);
    size_t per = (kHostPieceBytes + stream_bytes - 1) / stream_bytes;
    // the post-filter is one sequential wave per (stream, channel): a piece of a few LONG streams would run it as
    // a handful of waves, piece after piece -- keep at least 32 streams together so they filter side by side
    if (per < 32) per = 32;
    if ((nstreams + per - 1) / per > kHostMaxPieces) per = (nstreams + kHostMaxPieces - 1) / kHostMaxPieces;
    // ... and whole ROUNDS of it: the post-filter kernel keeps 4 workgroups = 8 chains per CU for the whole launch
    // (nyq_celt_post_round_chains), so a piece of a few chains more than a round runs a second, nearly empty round at the
    // price of a full one (1280 stereo streams: 1.75 ms, 1024: 1.0).  Pieces larger than a round are cut at multiples of it.
    const size_t round_streams = nyq_celt_post_round_chains(ctx) / (size_t)channels;
    if (round_streams >= 32 && per > round_streams) per = per / round_streams * round_streams;
    const size_t npieces = (nstreams + per - 1) / per;
    // FEW LONG streams (one piece): the call is cut in TIME instead -- windows of W frames (a multiple of 64: the kernels' in-wave
    // carry chains restart at the same frames as in one launch, bit-identical), upload of window w+1, kernels of window w and
    // download of window w-1 at the same time; the decoder states travel from window to window in device memory.  This is the
    // shape of a batch decoder's time slices when the GPU side is what the job waits for.
    if (npieces == 1 && nframes >= 128) {
        const size_t frame_in = bytes ? sym_bytes(channels, LM) : sym ? rec : (size_t)channels * N * sizeof(float);
        size_t W = ((kHostWindowBytes / (nstreams * frame_in)) / 64) * 64;
        if (W < 64) W = 64;
        if (ctx->opt_host_window > 0) W = (size_t)ctx->opt_host_window;  // NYQ_OPT_HOST_WINDOW
        const size_t nwin = (nframes + W - 1) / W;
        if (nwin >= 2 && nwin <= kHostMaxWindows) {
            rc = need_copy_streams(ctx, 2 * nwin + 2);
            if (rc != NYQ_OK) return rc;
            DrainOnExit drain{ctx};
            hipStream_t hs = ctx->s_h2d, ds = ctx->s_d2h;
            hipEvent_t ev0 = ctx->ev_pool[2 * nwin];
            NYQ_HIP(ctx, hipEventRecord(ev0, ctx->stream));
            NYQ_HIP(ctx, hipStreamWaitEvent(hs, ev0, 0));
            NYQ_HIP(ctx, hipStreamWaitEvent(ds, ev0, 0));
            // per-frame parameters, descriptors and the states: small, the whole call's at once
            if (!bytes) {
                if (transient) NYQ_HIP(ctx, copy_rows(d_t, nframes, transient, hstride, nframes, nstreams, hipMemcpyHostToDevice, hs));
                NYQ_HIP(ctx, copy_rows(d_pg, nframes * 4, pf_gain, hstride * 4, nframes * 4, nstreams, hipMemcpyHostToDevice, hs));
                NYQ_HIP(ctx, copy_rows(d_pp, nframes * 4, pf_pitch, hstride * 4, nframes * 4, nstreams, hipMemcpyHostToDevice, hs));
                NYQ_HIP(ctx, copy_rows(d_pt, nframes * 4, pf_tapset, hstride * 4, nframes * 4, nstreams, hipMemcpyHostToDevice, hs));
            } else {
                // the bytes form: every frame's bytes (a fortieth of the records they become) go up at once and the entropy stage
                // runs for the whole call before the first window; the windows then overlap shapes + chain with the downloads
                NYQ_HIP(ctx, up_bytes(0, nstreams, hs, h_es));
                hipEvent_t ev_ent = ctx->ev_pool[2 * nwin + 1];
                NYQ_HIP(ctx, hipEventRecord(ev_ent, hs));
                NYQ_HIP(ctx, hipStreamWaitEvent(ctx->stream, ev_ent, 0));
                if ((rc = run_entropy(0, nstreams)) != NYQ_OK) return rc;
            }
            // a window's output records: the stream's with t0 moved to the window's first sample (need_scratch above sized d_ds
            // for one set; the windows' sets live behind the symbol region: see n_dw)
            std::vector<nyq_out_desc> wdesc;
            nyq_out_desc *d_dw = nullptr;
            if (desc) {
                wdesc.resize(nwin * nstreams);
                for (size_t w = 0; w < nwin; w++)
                    for (size_t k = 0; k < nstreams; k++) {
                        wdesc[w * nstreams + k] = desc[k];
                        wdesc[w * nstreams + k].t0 += (long long)(w * W * N);
                    }
                d_dw = reinterpret_cast<nyq_out_desc *>(d_pfo + n_pf + n_ds + n_sy);
                NYQ_HIP(ctx, hipMemcpyAsync(d_dw, wdesc.data(), wdesc.size() * sizeof(nyq_out_desc), hipMemcpyHostToDevice, hs));
            }
            if (offsets) NYQ_HIP(ctx, hipMemcpyAsync(d_of, h_off.data(), h_off.size() * sizeof(unsigned), hipMemcpyHostToDevice, hs));
            if (state) {
                NYQ_HIP(ctx, hipMemcpyAsync(d_ov, h_ov, nsc * NYQ_HALF_OV * sizeof(float), hipMemcpyHostToDevice, hs));
                NYQ_HIP(ctx, hipMemcpyAsync(d_hi, h_hi, nsc * kPostHist * sizeof(float), hipMemcpyHostToDevice, hs));
                NYQ_HIP(ctx, hipMemcpyAsync(d_de, h_de, nsc * sizeof(float), hipMemcpyHostToDevice, hs));
                NYQ_HIP(ctx, hipMemcpyAsync(d_pfi, h_pf, nstreams * 6 * sizeof(float), hipMemcpyHostToDevice, hs));
            } else {                                                 // a fresh decoder: zero state, carried between the windows all the same
                NYQ_HIP(ctx, hipMemsetAsync(d_ov, 0, nsc * NYQ_HALF_OV * sizeof(float), hs));
                NYQ_HIP(ctx, hipMemsetAsync(d_hi, 0, nsc * kPostHist * sizeof(float), hs));
                NYQ_HIP(ctx, hipMemsetAsync(d_de, 0, nsc * sizeof(float), hs));
                NYQ_HIP(ctx, hipMemsetAsync(d_pfi, 0, nstreams * 6 * sizeof(float), hs));
            }
            float *pf_a = d_pfi, *pf_b = d_pfo;
            for (size_t w = 0; w < nwin; w++) {
                const size_t f0 = w * W, len = nframes - f0 < W ? nframes - f0 : W;
                hipEvent_t up = ctx->ev_pool[2 * w], done = ctx->ev_pool[2 * w + 1];
                if (bytes) {}
                else if (sym) NYQ_HIP(ctx, up_sym(0, nstreams, f0, len, hs));
                else NYQ_HIP(ctx, copy_rows(d_x + f0 * channels * N, nframes * channels * N * sizeof(float), freq + f0 * channels * N,
                                            hstride * channels * N * sizeof(float), len * channels * N * sizeof(float), nstreams, hipMemcpyHostToDevice, hs));
                NYQ_HIP(ctx, hipEventRecord(up, hs));
                NYQ_HIP(ctx, hipStreamWaitEvent(ctx->stream, up, 0));
                if (sym) rc = shape_core(ctx, offsets ? d_sy : d_sy + f0 * rec, d_x + f0 * channels * N, nstreams, len, channels, nframes, nframes,
                                         offsets ? d_of + f0 : nullptr, ostr, LM, rec);
                if (rc == NYQ_OK)
                    rc = chain_core(ctx, LM, d_x + f0 * channels * N, (transient || bytes) ? d_t + f0 : nullptr, d_pp + f0, d_pg + f0, d_pt + f0, pf_a, pf_b, d_ov,
                                    d_hi, d_de, d_out + f0 * N * channels, d_pcm, d_w, nstreams, len, channels, nframes,
                                    reinterpret_cast<const OutDesc *>(desc ? d_dw + w * nstreams : nullptr));
                if (rc != NYQ_OK) return rc;                         // (DrainOnExit waits for what is in flight)
                NYQ_HIP(ctx, hipEventRecord(done, ctx->stream));
                NYQ_HIP(ctx, hipStreamWaitEvent(ds, done, 0));
                if (!all_mapped)
                    NYQ_HIP(ctx, copy_rows(out + f0 * N * channels, hstride * channels * N * sizeof(float), d_out + f0 * N * channels,
                                           nframes * channels * N * sizeof(float), len * channels * N * sizeof(float), nstreams, hipMemcpyDeviceToHost, ds));
                float *t = pf_a;
                pf_a = pf_b;
                pf_b = t;
            }
            if (state) {                                             // (ds is behind the last window's kernels)
                NYQ_HIP(ctx, hipMemcpyAsync(h_ov, d_ov, nsc * NYQ_HALF_OV * sizeof(float), hipMemcpyDeviceToHost, ds));
                NYQ_HIP(ctx, hipMemcpyAsync(h_hi, d_hi, nsc * kPostHist * sizeof(float), hipMemcpyDeviceToHost, ds));
                NYQ_HIP(ctx, hipMemcpyAsync(h_de, d_de, nsc * sizeof(float), hipMemcpyDeviceToHost, ds));
                NYQ_HIP(ctx, hipMemcpyAsync(h_pf, pf_a, nstreams * 6 * sizeof(float), hipMemcpyDeviceToHost, ds));
                if (bytes) NYQ_HIP(ctx, hipMemcpyAsync(h_es, d_es, nstreams * sizeof(EnergyState), hipMemcpyDeviceToHost, ds));
            }
            if ((rc = wait_blocking(ctx, ds)) != NYQ_OK) return rc;
            if ((rc = wait_blocking(ctx, ctx->stream)) != NYQ_OK) return rc;
            return NYQ_OK;
        }
    }
    rc = need_copy_streams(ctx, 2 * npieces + 1);
    if (rc != NYQ_OK) return rc;
    DrainOnExit drain{ctx};
    hipEvent_t ev0 = ctx->ev_pool[2 * npieces];
    NYQ_HIP(ctx, hipEventRecord(ev0, ctx->stream));
    NYQ_HIP(ctx, hipStreamWaitEvent(ctx->s_h2d, ev0, 0));
    NYQ_HIP(ctx, hipStreamWaitEvent(ctx->s_d2h, ev0, 0));
    const size_t work_per_stream = (size_t)channels * (nframes + 1) * NYQ_HALF_OV;   // nyq_celt_synth_work_floats layout
    for (size_t k = 0; k < npieces; k++) {
        const size_t s0 = k * per, cnt = (nstreams - s0 < per ? nstreams - s0 : per);
        const size_t xo = s0 * nframes * channels * N, xn = cnt * nframes * channels * N;   // freq / pcm / out floats
        const size_t fo = s0 * nframes, fn = cnt * nframes;                                   // per-frame parameters
        const size_t co = s0 * channels, cn = cnt * (size_t)channels;                         // (stream, channel) units
        hipEvent_t up = ctx->ev_pool[2 * k], done = ctx->ev_pool[2 * k + 1];
        hipStream_t hs = ctx->s_h2d, ds = ctx->s_d2h;
        const size_t hx = s0 * hstride * channels * N, hf = s0 * hstride;            // host offsets of stream s0
        const size_t xw = nframes * channels * N * sizeof(float), xp = hstride * channels * N * sizeof(float);
        (void)xn;
        (void)fn;
        if (bytes) {
            NYQ_HIP(ctx, up_bytes(s0, cnt, hs, h_es));
        } else if (sym) {
            if (offsets && k == 0) NYQ_HIP(ctx, hipMemcpyAsync(d_of, h_off.data(), h_off.size() * sizeof(unsigned), hipMemcpyHostToDevice, hs));
            NYQ_HIP(ctx, up_sym(s0, cnt, 0, nframes, hs));
        } else {
            NYQ_HIP(ctx, copy_rows(d_x + xo, xw, freq + hx, xp, xw, cnt, hipMemcpyHostToDevice, hs));
        }
        if (!bytes) {
            if (transient) NYQ_HIP(ctx, copy_rows(d_t + fo, nframes, transient + hf, hstride, nframes, cnt, hipMemcpyHostToDevice, hs));
            NYQ_HIP(ctx, copy_rows(d_pg + fo, nframes * 4, pf_gain + hf, hstride * 4, nframes * 4, cnt, hipMemcpyHostToDevice, hs));
            NYQ_HIP(ctx, copy_rows(d_pp + fo, nframes * 4, pf_pitch + hf, hstride * 4, nframes * 4, cnt, hipMemcpyHostToDevice, hs));
            NYQ_HIP(ctx, copy_rows(d_pt + fo, nframes * 4, pf_tapset + hf, hstride * 4, nframes * 4, cnt, hipMemcpyHostToDevice, hs));
        }
        if (desc) NYQ_HIP(ctx, hipMemcpyAsync(d_ds + s0, desc + s0, cnt * sizeof(nyq_out_desc), hipMemcpyHostToDevice, hs));
        if (state) {
            NYQ_HIP(ctx, hipMemcpyAsync(d_ov + co * NYQ_HALF_OV, h_ov + co * NYQ_HALF_OV, cn * NYQ_HALF_OV * sizeof(float), hipMemcpyHostToDevice, hs));
            NYQ_HIP(ctx, hipMemcpyAsync(d_hi + co * kPostHist, h_hi + co * kPostHist, cn * kPostHist * sizeof(float), hipMemcpyHostToDevice, hs));
            NYQ_HIP(ctx, hipMemcpyAsync(d_de + co, h_de + co, cn * sizeof(float), hipMemcpyHostToDevice, hs));
            NYQ_HIP(ctx, hipMemcpyAsync(d_pfi + s0 * 6, h_pf + s0 * 6, cnt * 6 * sizeof(float), hipMemcpyHostToDevice, hs));
        }
        NYQ_HIP(ctx, hipEventRecord(up, hs));
        NYQ_HIP(ctx, hipStreamWaitEvent(ctx->stream, up, 0));
        if (bytes && (rc = run_entropy(s0, cnt)) != NYQ_OK) {
            (void)hipStreamSynchronize(hs);
            (void)hipStreamSynchronize(ctx->stream);
            (void)hipStreamSynchronize(ds);
            return rc;
        }
        if (sym && (rc = shape_core(ctx, d_sy + fo * rec, d_x + xo, cnt, nframes, channels, nframes, 0, offsets ? d_of + s0 * ostr : nullptr, ostr, LM, rec)) != NYQ_OK) {
            (void)hipStreamSynchronize(hs);
            (void)hipStreamSynchronize(ctx->stream);
            (void)hipStreamSynchronize(ds);
            return rc;
        }
        // (one launch for 20 ms stereo frames, synthesis + post-filter through d_pcm otherwise)
        rc = chain_core(ctx, LM, d_x + xo, (transient || bytes) ? d_t + fo : nullptr, d_pp + fo, d_pg + fo, d_pt + fo,
                        state ? d_pfi + s0 * 6 : nullptr, state ? d_pfo + s0 * 6 : nullptr,
                        state ? d_ov + co * NYQ_HALF_OV : nullptr, state ? d_hi + co * kPostHist : nullptr,
                        state ? d_de + co : nullptr, d_out + xo, d_pcm + xo, d_w + s0 * work_per_stream, cnt, nframes, channels, 0,
                        reinterpret_cast<const OutDesc *>(desc ? d_ds + s0 : nullptr));
        if (rc != NYQ_OK) {
            (void)hipStreamSynchronize(hs);
            (void)hipStreamSynchronize(ctx->stream);
            (void)hipStreamSynchronize(ds);
            return rc;
        }
        NYQ_HIP(ctx, hipEventRecord(done, ctx->stream));
        NYQ_HIP(ctx, hipStreamWaitEvent(ds, done, 0));
        if (!all_mapped) NYQ_HIP(ctx, copy_rows(out + hx, xp, d_out + xo, xw, xw, cnt, hipMemcpyDeviceToHost, ds));
        if (state) {
            NYQ_HIP(ctx, hipMemcpyAsync(h_ov + co * NYQ_HALF_OV, d_ov + co * NYQ_HALF_OV, cn * NYQ_HALF_OV * sizeof(float), hipMemcpyDeviceToHost, ds));
            NYQ_HIP(ctx, hipMemcpyAsync(h_hi + co * kPostHist, d_hi + co * kPostHist, cn * kPostHist * sizeof(float), hipMemcpyDeviceToHost, ds));
            NYQ_HIP(ctx, hipMemcpyAsync(h_de + co, d_de + co, cn * sizeof(float), hipMemcpyDeviceToHost, ds));
            NYQ_HIP(ctx, hipMemcpyAsync(h_pf + s0 * 6, d_pfo + s0 * 6, cnt * 6 * sizeof(float), hipMemcpyDeviceToHost, ds));
            if (bytes) NYQ_HIP(ctx, hipMemcpyAsync(h_es + s0 * kEsFloats, d_es + s0, cnt * sizeof(EnergyState), hipMemcpyDeviceToHost, ds));
        }
    }
    if ((rc = wait_blocking(ctx, ctx->s_d2h)) != NYQ_OK) return rc;
    if ((rc = wait_blocking(ctx, ctx->stream)) != NYQ_OK) return rc;
    return NYQ_OK;
}

extern "C" int nyq_celt_frames_to_pcm(nyq_ctx *ctx, int LM, const float *freq, const unsigned char *transient,
                                      const int *pf_pitch, const float *pf_gain, const int *pf_tapset, float *out,
                                      float *state, size_t nstreams, size_t nframes, int channels) {
    return frames_to_pcm_core(ctx, LM, freq, transient, pf_pitch, pf_gain, pf_tapset, out, state, nstreams, nframes, channels, nframes);
}

extern "C" int nyq_celt_frames_to_pcm_mapped(nyq_ctx *ctx, int LM, const float *freq, const unsigned char *transient,
                                             const int *pf_pitch, const float *pf_gain, const int *pf_tapset, float *out,
                                             const nyq_out_desc *desc, float *state, size_t nstreams, size_t nframes, int channels,
                                             size_t frames_per_stream) {
    if (ctx && frames_per_stream < nframes)
        return fail(ctx, NYQ_ERR_INVALID, "nyq_celt_frames_to_pcm_mapped: frames_per_stream is smaller than nframes");
    return frames_to_pcm_core(ctx, LM, freq, transient, pf_pitch, pf_gain, pf_tapset, out, state, nstreams, nframes, channels,
                              frames_per_stream, desc);
}

extern "C" int nyq_celt_symbols_to_pcm_mapped(nyq_ctx *ctx, int LM, const void *sym, const unsigned char *transient, const int *pf_pitch,
                                              const float *pf_gain, const int *pf_tapset, float *out, const nyq_out_desc *desc,
                                              float *state, size_t nstreams, size_t nframes, int channels, size_t frames_per_stream) {
    return nyq_celt_symbols_packed_to_pcm_mapped(ctx, LM, sym, nullptr, 0, transient, pf_pitch, pf_gain, pf_tapset, out, desc, state, nstreams,
                                                 nframes, channels, frames_per_stream);
}

extern "C" int nyq_celt_symbols_packed_to_pcm_mapped(nyq_ctx *ctx, int LM, const void *sym, const unsigned *offsets, size_t stream_bytes,
                                                     const unsigned char *transient, const int *pf_pitch, const float *pf_gain,
                                                     const int *pf_tapset, float *out, const nyq_out_desc *desc, float *state,
                                                     size_t nstreams, size_t nframes, int channels, size_t frames_per_stream) {
    if (ctx && frames_per_stream < nframes)
        return fail(ctx, NYQ_ERR_INVALID, "nyq_celt_symbols_to_pcm_mapped: frames_per_stream is smaller than nframes");
    if (ctx && !sym) return fail(ctx, NYQ_ERR_INVALID, "nyq_celt_symbols_to_pcm_mapped: NULL buffer");
    return frames_to_pcm_core(ctx, LM, nullptr, transient, pf_pitch, pf_gain, pf_tapset, out, state, nstreams, nframes, channels,
                              frames_per_stream, desc, static_cast<const unsigned char *>(sym), offsets, stream_bytes);
}

extern "C" int nyq_celt_bytes_to_pcm_mapped(nyq_ctx *ctx, int LM, const unsigned char *bytes, const unsigned *frame_words, float *out,
                                            const nyq_out_desc *desc, float *state, size_t nstreams, size_t nframes, int channels,
                                            size_t frames_per_stream) {
    if (ctx && frames_per_stream < nframes)
        return fail(ctx, NYQ_ERR_INVALID, "nyq_celt_bytes_to_pcm_mapped: frames_per_stream is smaller than nframes");
    if (ctx && (!bytes || !frame_words)) return fail(ctx, NYQ_ERR_INVALID, "nyq_celt_bytes_to_pcm_mapped: NULL buffer");
    if (ctx && channels != 1 && channels != 2) return fail(ctx, NYQ_ERR_INVALID, "nyq_celt_bytes_to_pcm_mapped: mono and stereo streams");
    return frames_to_pcm_core(ctx, LM, nullptr, nullptr, nullptr, nullptr, nullptr, out, state, nstreams, nframes, channels, frames_per_stream, desc,
                              bytes, nullptr, 0, frame_words);
}
extern "C" size_t nyq_celt_byte_slot(void) { return kByteSlot; }

extern "C" int nyq_celt_frames_to_pcm_window(nyq_ctx *ctx, int LM, const float *freq, const unsigned char *transient,
                                             const int *pf_pitch, const float *pf_gain, const int *pf_tapset, float *out,
                                             float *state, size_t nstreams, size_t nframes, int channels,
                                             size_t frames_per_stream) {
    if (ctx && frames_per_stream < nframes)
        return fail(ctx, NYQ_ERR_INVALID, "nyq_celt_frames_to_pcm_window: frames_per_stream is smaller than nframes");
    return frames_to_pcm_core(ctx, LM, freq, transient, pf_pitch, pf_gain, pf_tapset, out, state, nstreams, nframes, channels,
                              frames_per_stream);
}

// ---- device copy: a measurement utility -----------------------------------------------------------
// How fast does THIS box stream N bytes in + N bytes out?  The row kernels read and write the same number of bytes, so a
// tuned plain copy is their practical ceiling; bench.py times these forms in-process and reports the best
// as roofline.measured_device_copy_GBps.  (tools/copybench.hip is the stand-alone survey the forms were picked from.)
template <int U>
__global__ __launch_bounds__(256) void copy_gs_kernel(const vf4 *__restrict__ in, vf4 *__restrict__ out, size_t n4) {
    const size_t stride = (size_t)gridDim.x * blockDim.x;
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    for (; i + (U - 1) * stride < n4; i += U * stride) {
        vf4 r[U];
#pragma unroll
        for (int k = 0; k < U; k++) r[k] = in[i + k * stride];
#pragma unroll
        for (int k = 0; k < U; k++) out[i + k * stride] = r[k];
    }
    for (; i < n4; i += stride) out[i] = in[i];
}
template <int CH, int NT>
__global__ __launch_bounds__(128) void copy_chunk_kernel(const vf4 *__restrict__ in, vf4 *__restrict__ out, size_t n4) {
    // each wave owns contiguous chunks of CH KB, loads a whole chunk into registers, then stores it (the row kernels' shape)
    const int lane = threadIdx.x & 63;
    const size_t wave = ((size_t)blockIdx.x * blockDim.x + threadIdx.x) >> 6, nw = ((size_t)gridDim.x * blockDim.x) >> 6;
    const size_t nchunks = n4 / (CH * 64);
    for (size_t c = wave; c < nchunks; c += nw) {
        const vf4 *p = in + c * (CH * 64) + lane;
        vf4 *q = out + c * (CH * 64) + lane;
        vf4 r[CH];
#pragma unroll
        for (int k = 0; k < CH; k++) r[k] = NT ? __builtin_nontemporal_load(p + 64 * k) : p[64 * k];
#pragma unroll
        for (int k = 0; k < CH; k++) {
            if (NT) __builtin_nontemporal_store(r[k], q + 64 * k);
            else q[64 * k] = r[k];
        }
    }
    for (size_t i = nchunks * (CH * 64) + wave * 64 + lane; i < n4; i += nw * 64) out[i] = in[i];
}

static const char *const kCopyForms[] = {"grid-stride float4, 8 blocks/CU", "grid-stride float4 x4, 4 blocks/CU", "grid-stride float4 x4, 6 blocks/CU",
                                         "15 KB chunk per wave, 6 waves/CU", "15 KB chunk per wave, 4 waves/CU", "4 KB chunk per wave, 6 waves/CU",
                                         "15 KB chunk per wave, non-temporal, 6 waves/CU"};
extern "C" int nyq_device_copy_forms(void) { return (int)(sizeof kCopyForms / sizeof kCopyForms[0]); }
extern "C" const char *nyq_device_copy_form_name(int form) {
    return form >= 0 && form < nyq_device_copy_forms() ? kCopyForms[form] : "";
}
extern "C" int nyq_device_copy_dev(nyq_ctx *ctx, void *d_dst, const void *d_src, size_t bytes, int form) {
    if (!ctx) return fail(nullptr, NYQ_ERR_INVALID, "nyq_device_copy_dev: ctx is NULL");
    if (!d_dst || !d_src || !aligned16(d_dst) || !aligned16(d_src) || (bytes & 15))
        return fail(ctx, NYQ_ERR_INVALID, "nyq_device_copy_dev: 16-byte aligned buffers and a multiple of 16 bytes");
    if (form < 0 || form >= nyq_device_copy_forms()) return fail(ctx, NYQ_ERR_INVALID, "nyq_device_copy_dev: unknown form");
    const size_t n4 = bytes / 16;
    if (n4 == 0) return NYQ_OK;
    const vf4 *in = static_cast<const vf4 *>(d_src);
    vf4 *out = static_cast<vf4 *>(d_dst);
    const unsigned cus = (unsigned)ctx->cus;
    switch (form) {
    case 0: hipLaunchKernelGGL(copy_gs_kernel<1>, dim3(cus * 8), dim3(256), 0, ctx->stream, in, out, n4); break;
    case 1: hipLaunchKernelGGL(copy_gs_kernel<4>, dim3(cus * 4), dim3(256), 0, ctx->stream, in, out, n4); break;
    case 2: hipLaunchKernelGGL(copy_gs_kernel<4>, dim3(cus * 6), dim3(256), 0, ctx->stream, in, out, n4); break;
    case 3: hipLaunchKernelGGL((copy_chunk_kernel<15, 0>), dim3(cus * 3), dim3(128), 0, ctx->stream, in, out, n4); break;
    case 4: hipLaunchKernelGGL((copy_chunk_kernel<15, 0>), dim3(cus * 2), dim3(128), 0, ctx->stream, in, out, n4); break;
    case 5: hipLaunchKernelGGL((copy_chunk_kernel<4, 0>), dim3(cus * 3), dim3(128), 0, ctx->stream, in, out, n4); break;
    default: hipLaunchKernelGGL((copy_chunk_kernel<15, 1>), dim3(cus * 3), dim3(128), 0, ctx->stream, in, out, n4); break;
    }
    NYQ_HIP(ctx, hipGetLastError());
    return NYQ_OK;
}

// ---- the reference's operator names (cuda/mdct_cuda.hpp:79-103) --------------------
// State of the drop-in entry points.  The reference keeps an unsynchronised global map keyed by a sum of buffer sizes
// (mdct_cuda.cu:558-584); here one mutex serialises the calls (the interface is synchronous and moves ~8 KB per call:
// there is nothing to win from concurrency inside it, and a decoder thread per stream simply takes turns), and the
// uploaded tables are identified by CONTENT -- a caller that refills the same buffer gets its new tables.
static std::mutex g_shim_mu;
static nyq_ctx *g_shim_ctx = nullptr;
static bool g_shim_tables_set = false;

// What the void entry points do when they cannot do their work.  Default: the reference's behaviour on device failure,
// fprintf + end of the process (mdct_cuda.cu:11-19 calls exit(1); here abort(), so that a core / a test log shows where).
// An integrator who would rather degrade -- log, mark the stream as failed, unwind -- installs a handler with
// nyq_shim_set_error_handler: it is called with the entry point's name and the reason, and if it returns, the call returns
// with `output` untouched (the carry in output[0 .. overlap/2) and everything behind it as the caller left them).  There is
// still no CPU fallback in this library: what to decode instead is the integrator's decision.
static void (*g_shim_handler)(const char *who, const char *what) = nullptr;

extern "C" void nyq_shim_set_error_handler(void (*handler)(const char *who, const char *what)) {
    std::lock_guard<std::mutex> lk(g_shim_mu);
    g_shim_handler = handler;
}

// A failed shim call: the reason, recorded under the lock; the handler runs AFTER the lock is released (it may call any
// entry point of this library, the shims and cleanupCudaBuffers included), and nothing is thrown across the foreign callback.
struct ShimFault {
    bool failed = false;
    bool hip = false;                 // a HIP failure: the shim's context is dropped so that the next call starts afresh
    std::string what;
    void set(const std::string &w, bool from_hip = false) {
        if (!failed) { failed = true; hip = from_hip; what = w; }
    }
};

static void shim_drop_ctx() {                                  // call with g_shim_mu held
    if (g_shim_ctx) nyq_ctx_destroy(g_shim_ctx);
    g_shim_ctx = nullptr;
    g_shim_tables_set = false;
}

static nyq_ctx *shim_ctx(ShimFault &F, const float *trig, const float *window) {   // call with g_shim_mu held
    if (!g_shim_ctx) {
        const char *dev = std::getenv("NYQ_DEVICE");
        if (nyq_ctx_create(&g_shim_ctx, dev ? std::atoi(dev) : 0) != NYQ_OK) {
            F.set(nyq_last_error(nullptr));
            g_shim_ctx = nullptr;
            return nullptr;
        }
        g_shim_tables_set = false;
    }
    // the reference uploads trig/window once per state (mdct_cuda.cu:577-579); here whenever their CONTENT differs from
    // what the context holds (2.4 KB compared per call)
    if (!g_shim_tables_set || std::memcmp(g_shim_ctx->h_trig, trig, sizeof g_shim_ctx->h_trig) != 0 ||
        std::memcmp(g_shim_ctx->h_window, window, sizeof g_shim_ctx->h_window) != 0) {
        if (nyq_ctx_set_tables(g_shim_ctx, trig, window) != NYQ_OK) {
            F.set(nyq_last_error(g_shim_ctx), true);
            return nullptr;
        }
        g_shim_tables_set = true;
    }
    return g_shim_ctx;
}

// One call of the reference's offload interface = one, two or eight rows.  The round trip is latency, not
// bandwidth, so the rows go through a small page-locked block that the GPU reads and writes in place over
// PCIe (host memory from hipHostMalloc is device-addressable): marshal -> ONE kernel launch -> synchronise
// -> copy out.  No H2D/D2H copy commands, no events.
constexpr int kShimRows = 8;
struct ShimPinned {
    float in[kShimRows][NYQ_MDCT_N / 2];
    float fin[kShimRows][NYQ_MDCT_N / 2];
    float carry[kShimRows][NYQ_HALF_OV + 4];   // (rows of `nch` calls are packed 60 floats apart from the start of the block)
    float tail[kShimRows][NYQ_HALF_OV + 4];
};
static ShimPinned *g_shim_pin = nullptr;

static void shim_rows_locked(ShimFault &F, int nch, const float *const *input, float *const *output, const float *trig,
                             int N, int shift, int stride, int overlap, const float *window) {
    if (shift < 0 || shift > 3 || N != (NYQ_MDCT_N >> shift) || overlap != NYQ_OVERLAP || stride < 1 || !trig || !window)
        return F.set("unsupported call: only the static 48 kHz mode (mdct.n 1920, overlap 120, shift 0..3) exists");
    for (int c = 0; c < nch; c++)
        if (!input[c] || !output[c]) return F.set("NULL input / output row");
    nyq_ctx *ctx = shim_ctx(F, trig, window);
    if (!ctx) return;
    if (!g_shim_pin) {
        g_shim_pin = static_cast<ShimPinned *>(nyq_host_alloc(sizeof(ShimPinned)));
        if (!g_shim_pin) return F.set("cannot allocate page-locked staging memory");
    }
    ShimPinned &P = *g_shim_pin;
    const int n2 = N >> 1;
    // rows must be contiguous per array: channel c + 1 right after channel c (n2 floats apart; carries 60 apart)
    float *pin = &P.in[0][0], *pfin = &P.fin[0][0], *pcar = &P.carry[0][0], *ptail = &P.tail[0][0];
    for (int c = 0; c < nch; c++) {
        const float *src = input[c];
        float *dst = pin + c * n2;
        if (stride == 1) std::memcpy(dst, src, sizeof(float) * n2);
        else for (int k = 0; k < n2; k++) dst[k] = src[(size_t)k * stride];   // argument marshalling
        std::memcpy(pcar + c * NYQ_HALF_OV, output[c], sizeof(float) * NYQ_HALF_OV);
    }
    if (nyq_imdct_batch_dev(ctx, shift, pin, pcar, pfin, ptail, (size_t)nch) != NYQ_OK || nyq_ctx_synchronize(ctx) != NYQ_OK)
        return F.set(nyq_last_error(ctx), true);
    for (int c = 0; c < nch; c++) {
        std::memcpy(output[c], pfin + c * n2, sizeof(float) * n2);
        std::memcpy(output[c] + n2, ptail + c * NYQ_HALF_OV, sizeof(float) * NYQ_HALF_OV);
    }
}

static void shim_rows(const char *who, int nch, const float *const *input, float *const *output, const float *trig,
                      int N, int shift, int stride, int overlap, const float *window) {
    ShimFault F;
    void (*handler)(const char *, const char *) = nullptr;
    {
        std::lock_guard<std::mutex> lk(g_shim_mu);
        shim_rows_locked(F, nch, input, output, trig, N, shift, stride, overlap, window);
        if (F.failed && F.hip) shim_drop_ctx();      // (a context that has seen a HIP error is not reused)
        handler = g_shim_handler;
    }
    if (!F.failed) return;
    if (handler) {
        handler(who, F.what.c_str());                // lock released: the handler may call back into the library
        return;                                      // `output` is untouched
    }
    std::fprintf(stderr, "%s: %s\n", who, F.what.c_str());
    std::abort();
}

extern "C" void processMDCTCuda(const float *input, float *output, const float *trig, int N, int shift, int stride,
                                float sine, int overlap, const float *window) {
    (void)sine;   // a function of N alone (mdct.c:292); the kernels carry it as a constant
    const float *in[1] = {input};
    float *out[1] = {output};
    shim_rows("processMDCTCuda", 1, in, out, trig, N, shift, stride, overlap, window);
}

extern "C" void processMDCTCudaB1C2(const float *input[2], float *output[2], const float *trig, int N, int shift,
                                    int stride, float sine, int overlap, const float *window) {
    (void)sine;
    shim_rows("processMDCTCudaB1C2", 2, input, output, trig, N, shift, stride, overlap, window);
}

// cuda/mdct_cuda.hpp:96-98.  Declared with eight row pointers; the reference never calls it and its own definition
// (mdct_cuda_b8.cu:482-501) takes two -- here it is what the declaration says: eight clt_mdct_backward rows of one size in
// one launch (= four B1C2 calls), so that a program linking the header's whole symbol set links and gets defined results.
extern "C" void processMDCTCudaB8C2(const float *input[8], float *output[8], const float *trig, int N, int shift,
                                    int stride, float sine, int overlap, const float *window) {
    (void)sine;
    shim_rows("processMDCTCudaB8C2", 8, input, output, trig, N, shift, stride, overlap, window);
}

extern "C" void cleanupCudaBuffers(void) {
    std::lock_guard<std::mutex> lk(g_shim_mu);
    if (g_shim_pin) nyq_host_free(g_shim_pin);
    g_shim_pin = nullptr;
    shim_drop_ctx();
}

extern "C" void printCudaVersion(void) {
    int rt = 0, drv = 0, ndev = 0;
    (void)hipRuntimeGetVersion(&rt);
    (void)hipDriverGetVersion(&drv);
    (void)hipGetDeviceCount(&ndev);
    std::printf("HIP runtime %d, driver %d, %d device(s) [libnyq_imdct, gfx950]\n", rt, drv, ndev);
}

#ifdef NYQ_PIPE_STAMPS
// diagnostic build only (tools/chain_stamps.py): the one-launch chain kernel's per-role cycle sums; reset != 0 clears them
extern "C" int nyq_debug_chain_stamps(unsigned long long *out32, int reset) {
    unsigned long long z[32] = {0};
    if (out32 && hipMemcpyFromSymbol(out32, HIP_SYMBOL(nyq::g_chain_stamps), sizeof z) != hipSuccess) return 1;
    if (reset && hipMemcpyToSymbol(HIP_SYMBOL(nyq::g_chain_stamps), z, sizeof z) != hipSuccess) return 1;
    return 0;
}
// diagnostic build only (tools/placement_trace.py): the post-filter pipeline's per-workgroup placement trace
extern "C" int nyq_debug_pipe_wg(void *out, size_t bytes) {
    return hipMemcpyFromSymbol(out, HIP_SYMBOL(nyq::g_pipe_wg), bytes < sizeof(nyq::g_pipe_wg) ? bytes : sizeof(nyq::g_pipe_wg)) == hipSuccess ? 0 : 1;
}
#endif
