// tests/sched/fake_gpu.cpp -- TEST ONLY.  A CPU stand-in for the few libnyq_imdct.so entry points the batch
// decoder calls, so that its scheduler (decode threads, feeder threads, time slices, ordered hand-over of the
// samples, sub-batches) can run under ThreadSanitizer where there is no GPU.  It does NOT decode audio: the
// "PCM" of a frame is a checksum-like function of that frame's coefficients, its parameters, and a running
// per-(stream, channel) state that is carried exactly like the real decoder state -- so any slice delivered out
// of order, twice, or with the wrong state changes the output.  Never linked into the product.
#include <atomic>
#include <chrono>
#include <cstdlib>
#include <cstring>
#include <thread>

#include "../../include/nyq_imdct.h"

struct nyq_ctx {
    int device;
};

// The stand-in exposes kFakeDevices "GPUs": contexts of any other index are refused like a missing device, and
// every call is counted against its context's device (fake_gpu_calls) so that a test can see the sharding.
static constexpr int kFakeDevices = 4;
static std::atomic<long> g_calls[kFakeDevices];
static std::atomic<long> g_created{0}, g_destroyed{0}, g_live{0}, g_allCalls{0};
// fault injection: every g_failCallEvery-th GPU call / every g_failCreateEvery-th context creation fails (0 = never)
static std::atomic<long> g_failCallEvery{0}, g_failCreateEvery{0};
// contexts destroyed while some GPU call of the process was in flight (the lease pool promises none: nqr_surface.cpp)
static std::atomic<long> g_inFlight{0}, g_destroyedBesideCalls{0};
struct InFlight {
    InFlight() { g_inFlight++; }
    ~InFlight() { g_inFlight--; }
};

extern "C" {

long fake_gpu_calls(int device) { return device >= 0 && device < kFakeDevices ? g_calls[device].load() : -1; }
void fake_gpu_counts(long *created, long *destroyed, long *live) { *created = g_created; *destroyed = g_destroyed; *live = g_live; }
void fake_gpu_fail_every(long calls, long creates) { g_failCallEvery = calls; g_failCreateEvery = creates; }
long fake_gpu_destroyed_beside_calls(void) { return g_destroyedBesideCalls.load(); }

int nyq_device_count(void) { return kFakeDevices; }

int nyq_ctx_create(nyq_ctx **out, int device) {
    if (device < 0 || device >= kFakeDevices) return NYQ_ERR_NO_DEVICE;
    static std::atomic<long> attempts{0};
    const long n = ++attempts, every = g_failCreateEvery.load();
    if (every > 0 && n % every == 0) return NYQ_ERR_HIP;           // an injected failure of context creation
    g_created++;
    g_live++;
    *out = new nyq_ctx{device};
    return NYQ_OK;
}
void nyq_ctx_destroy(nyq_ctx *c) {
    if (!c) return;
    if (g_inFlight.load() > 0) g_destroyedBesideCalls++;
    g_destroyed++;
    g_live--;
    delete c;
}
const char *nyq_last_error(const nyq_ctx *) { return "fake"; }
int nyq_ctx_set_option(nyq_ctx *, int, long) { return NYQ_OK; }
void *nyq_host_alloc(size_t bytes) { return std::malloc(bytes ? bytes : 1); }
void nyq_host_free(void *p) { std::free(p); }

size_t nyq_celt_state_floats(size_t nstreams, int channels) {
    const size_t nsc = nstreams * (size_t)(channels > 0 ? channels : 0);
    return nsc * (60 + 1088 + 1) + nstreams * 6;
}

// "device memory" of the stand-in is host memory
void *nyq_device_alloc(nyq_ctx *, size_t bytes) { return std::malloc(bytes ? bytes : 1); }
void nyq_device_free(nyq_ctx *, void *p) { std::free(p); }
int nyq_device_zero(nyq_ctx *, void *p, size_t bytes) { std::memset(p, 0, bytes); return NYQ_OK; }
int nyq_device_download(nyq_ctx *, void *dst, const void *src, size_t bytes) { std::memcpy(dst, src, bytes); return NYQ_OK; }
int nyq_device_dup_channel(nyq_ctx *, float *base, int cstride, int src, int dst, size_t n) {
    for (size_t t = 0; t < n; t++) base[t * cstride + dst] = base[t * cstride + src];
    return NYQ_OK;
}

// desc != null: stream s with desc[s].base writes through its record (as the real kernels' store phase does)
int nyq_celt_frames_to_pcm_mapped(nyq_ctx *ctx, int LM, const float *freq, const unsigned char *transient, const int *pf_pitch,
                                  const float *pf_gain, const int *pf_tapset, float *out, const nyq_out_desc *desc, float *state,
                                  size_t nstreams, size_t nframes, int channels, size_t frames_per_stream) {
    const size_t N = (size_t)120 << LM, nsc = nstreams * channels;
    InFlight guard;
    g_calls[ctx->device]++;
    const long ncall = ++g_allCalls, every = g_failCallEvery.load();
    if (every > 0 && ncall % every == 0) return NYQ_ERR_HIP;      // an injected device failure
    for (size_t s = 0; s < nstreams; s++)
        for (int c = 0; c < channels; c++) {
            // the running state lives where the real overlap state lives: first float of the stream-channel's 60
            float acc = state ? state[(s * channels + c) * 60] : 0.f;
            for (size_t f = 0; f < nframes; f++) {
                const size_t hf = s * frames_per_stream + f;
                const float *x = freq + (hf * channels + c) * N;
                float v = 0.f;
                for (size_t k = 0; k < N; k += 7) v += x[k];
                acc = 0.5f * acc + v + (transient ? (float)transient[hf] : 0.f) + (float)pf_pitch[hf] * 1e-3f + pf_gain[hf] +
                      (float)pf_tapset[hf];
                if (desc && desc[s].base) {
                    const nyq_out_desc &D = desc[s];
                    if (c < 2 && D.coff[c] >= 0)
                        for (size_t k = 0; k < N; k++) {
                            const long long ts = D.t0 + (long long)(f * N + k);
                            if (ts >= D.first && ts < D.last) D.base[(ts - D.first) * D.cstride + D.coff[c]] = (acc + (float)k * 1e-6f) * D.gain;
                        }
                } else {
                    float *o = out + (hf * N) * channels + c;
                    for (size_t k = 0; k < N; k++) o[k * channels] = acc + (float)k * 1e-6f;
                }
            }
            if (state) state[(s * channels + c) * 60] = acc;
        }
    if (state) {   // touch the rest of the state the way the real call does (read-modify-write)
        float *hi = state + nsc * 60, *pf = hi + nsc * 1088 + nsc;
        for (size_t i = 0; i < nsc * 1088; i += 64) hi[i] += 1.f;
        for (size_t s = 0; s < nstreams; s++) pf[s * 6] += (float)nframes;
    }
    std::this_thread::sleep_for(std::chrono::microseconds(200 + 20 * (nstreams * nframes) / 64));   // a GPU takes a while
    return NYQ_OK;
}

// Symbol records: the stand-in's "band shapes" are a digest of everything the record DEFINES (head, gains, operations, vectors,
// leaves, or the host-built freq[]) spread over a freq[] buffer, which then goes through the call above -- so a record that is
// stale, misplaced or half written changes the output.  (Bytes the entropy stage leaves undefined -- unused slots -- are not
// read, as the real kernel does not read them.)
size_t nyq_celt_symbol_bytes_lm(int channels, int LM) {
    if ((channels != 1 && channels != 2) || LM < 0 || LM > 3) return 0;
    const size_t full = (size_t)3072 + (size_t)channels * 3840, floor_ = LM == 3 ? 0 : (size_t)2048 * channels + 512;
    const size_t scaled = LM == 3 ? full : LM == 2 ? full * 5 / 8 : full >> (3 - LM);
    return ((scaled > floor_ ? scaled : floor_) + 15) & ~(size_t)15;
}
size_t nyq_celt_symbol_bytes(int channels) { return nyq_celt_symbol_bytes_lm(channels, 3); }
int nyq_celt_symbols_packed_to_pcm_mapped(nyq_ctx *ctx, int LM, const void *sym, const unsigned int *offsets, size_t stream_bytes,
                                          const unsigned char *transient, const int *pf_pitch, const float *pf_gain, const int *pf_tapset,
                                          float *out, const nyq_out_desc *desc, float *state, size_t nstreams, size_t nframes, int channels,
                                          size_t frames_per_stream) {
    const size_t rec = nyq_celt_symbol_bytes_lm(channels, LM), per = (size_t)channels * ((size_t)120 << LM);
    float *freq = (float *)std::malloc(sizeof(float) * nstreams * frames_per_stream * per);
    for (size_t s = 0; s < nstreams; s++)
        for (size_t f = 0; f < nframes; f++) {
            const unsigned char *r = offsets ? (const unsigned char *)sym + s * stream_bytes + (size_t)offsets[s * (frames_per_stream + 1) + f] * 16
                                             : (const unsigned char *)sym + (s * frames_per_stream + f) * rec;
            float *x = freq + (s * frames_per_stream + f) * per;
            const nyq_sym_head *H = (const nyq_sym_head *)r;
            if (H->flags & NYQ_SYM_HOST_FREQ) {
                std::memcpy(x, r + 32, sizeof(float) * per);
                continue;
            }
            unsigned h = 2166136261u;
            auto mix = [&](const void *p, size_t n) {
                for (size_t k = 0; k < n; k++) h = (h ^ ((const unsigned char *)p)[k]) * 16777619u;
            };
            mix(H, sizeof *H);
            if (H->nops) {
                for (int c = 0; c < H->channels; c++) mix(r + 32 + 4 * (c * 21 + H->start), 4 * (size_t)(H->end - H->start));
                mix(r + 200, sizeof(nyq_sym_op) * H->nops + sizeof(nyq_sym_vec) * H->nvecs + sizeof(nyq_sym_leaf) * H->nleaves);
            }
            for (size_t k = 0; k < per; k++) x[k] = H->nops ? (float)((h >> (k % 13)) & 0xff) * (1.f / 64) : 0.f;
        }
    const int rc = nyq_celt_frames_to_pcm_mapped(ctx, LM, freq, transient, pf_pitch, pf_gain, pf_tapset, out, desc, state, nstreams, nframes,
                                                 channels, frames_per_stream);
    std::free(freq);
    return rc;
}
int nyq_celt_symbols_to_pcm_mapped(nyq_ctx *ctx, int LM, const void *sym, const unsigned char *transient, const int *pf_pitch, const float *pf_gain,
                                   const int *pf_tapset, float *out, const nyq_out_desc *desc, float *state, size_t nstreams, size_t nframes,
                                   int channels, size_t frames_per_stream) {
    return nyq_celt_symbols_packed_to_pcm_mapped(ctx, LM, sym, nullptr, 0, transient, pf_pitch, pf_gain, pf_tapset, out, desc, state, nstreams,
                                                 nframes, channels, frames_per_stream);
}

// (the device's own entropy stage: the stand-in has none -- the scheduler tests run with NYQ_DEVICE_ENTROPY unset)
int nyq_ctx_set_entropy_tables(nyq_ctx *, const void *, size_t) { return NYQ_ERR_INVALID; }
size_t nyq_celt_byte_slot(void) { return 1280; }
int nyq_celt_bytes_to_pcm_mapped(nyq_ctx *, int, const unsigned char *, const unsigned int *, float *, const nyq_out_desc *, float *, size_t, size_t, int,
                                 size_t) {
    return NYQ_ERR_INVALID;
}

int nyq_celt_frames_to_pcm_window(nyq_ctx *ctx, int LM, const float *freq, const unsigned char *transient, const int *pf_pitch,
                                  const float *pf_gain, const int *pf_tapset, float *out, float *state, size_t nstreams,
                                  size_t nframes, int channels, size_t frames_per_stream) {
    return nyq_celt_frames_to_pcm_mapped(ctx, LM, freq, transient, pf_pitch, pf_gain, pf_tapset, out, nullptr, state, nstreams, nframes,
                                         channels, frames_per_stream);
}

int nyq_celt_frames_to_pcm(nyq_ctx *c, int LM, const float *freq, const unsigned char *transient, const int *pf_pitch,
                           const float *pf_gain, const int *pf_tapset, float *out, float *state, size_t nstreams, size_t nframes,
                           int channels) {
    return nyq_celt_frames_to_pcm_window(c, LM, freq, transient, pf_pitch, pf_gain, pf_tapset, out, state, nstreams, nframes, channels,
                                         nframes);
}

}  // extern "C"
