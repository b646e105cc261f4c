// nyq_host_capi.cpp -- plain-C entry points of libnyquist_host.so used by the tests (ctypes) and
// by tools: the CPU entropy stage on its own, and the mode tables for cross-checking.
#include <chrono>
#include <cstdlib>
#include <cstring>
#include <mutex>
#include <stdexcept>
#include <string>
#include <vector>

#include "../../include/nyq_imdct.h"
#include "batch_decoder.hpp"
#include "celt_decoder.hpp"
#include "libnyquist/Decoders.h"
#include "opus_stream.hpp"

using namespace nyq_host;

static thread_local std::string g_capi_err;

extern "C" {

// RangeDecoder::tellFrac's table form against the reference's squaring loop, every case (0 = identical)
long nyqh_tell_frac_self_check(void) { return RangeDecoder::tellFracSelfCheck(); }

// cache tables of the 48 kHz mode as this library computed them (tests compare them with the
// reference's static tables): returns sizes through the out parameters
int nyqh_mode_tables(short *logN21, short *cacheIndex105, unsigned char *cacheBits, int *cacheBitsLen,
                     unsigned char *cacheCaps168) {
    const CeltMode &m = mode48k();
    std::memcpy(logN21, m.logN, sizeof m.logN);
    if ((int)m.cacheIndex.size() != 105 || (int)m.cacheCaps.size() != 168) return -1;
    std::memcpy(cacheIndex105, m.cacheIndex.data(), 105 * sizeof(short));
    *cacheBitsLen = (int)m.cacheBits.size();
    std::memcpy(cacheBits, m.cacheBits.data(), m.cacheBits.size());
    std::memcpy(cacheCaps168, m.cacheCaps.data(), 168);
    return 0;
}

// Entropy-decode an Ogg Opus file held in memory up to freq[] (no GPU involved).
// Outputs for at most max_frames frames: freq [frames][channels][frame_size] (all frames must share
// one frame size), flags [frames][4] = {transient, pf_pitch, pf_tapset, LM}, pf_gain [frames],
// range [frames] (final range coder state per frame).  info[0..5] = channels, pre_skip, frames,
// frame_size, last_granule (low 32 bits), packets.
// Returns 0, or <0: -10 malformed container, -11 unsupported packet (not CELT-only / mapping family),
// -12 mixed frame sizes, otherwise the CELT decoder's error.
int nyqh_decode_to_freq(const unsigned char *file, long size, long max_frames, float *freq, int *flags, float *pf_gain,
                        unsigned *range, long *info) {
    OggOpusFile f;
    try {
        f = parseOggOpus(file, (size_t)size);
    } catch (const std::exception &) {
        return -10;
    }
    if (f.head.mappingFamily != 0 || f.head.channels < 1 || f.head.channels > 2) return -11;
    const int CC = f.head.channels;
    CeltDecoder dec(CC);
    long nframes = 0;
    int frameSize = 0;
    for (const auto &pkt : f.packets) {
        PacketFrames pf;
        if (!parseOpusPacket(pkt.data(), (int)pkt.size(), pf)) return -10;
        if (pf.config < 16) return -11;
        dec.setEndBand(pf.bandwidthEnd);
        dec.setStreamChannels(pf.stereo ? 2 : 1);
        for (const auto &fr : pf.frames) {
            if (nframes >= max_frames) goto done;
            if (frameSize == 0) frameSize = pf.frameSize;
            if (pf.frameSize != frameSize) goto done;   // a different frame size ends this fixed-shape dump
            CeltFrame info1;
            const int rc = dec.decode(fr.first, fr.second, pf.frameSize, freq + (size_t)nframes * CC * frameSize, info1);
            if (rc < 0) return rc;
            flags[4 * nframes + 0] = info1.transient;
            flags[4 * nframes + 1] = info1.pfPitch;
            flags[4 * nframes + 2] = info1.pfTapset;
            flags[4 * nframes + 3] = info1.LM;
            pf_gain[nframes] = info1.pfGain;
            range[nframes] = info1.rangeFinal;
            nframes++;
        }
    }
done:
    info[0] = CC;
    info[1] = f.head.preSkip;
    info[2] = nframes;
    info[3] = frameSize;
    info[4] = (long)f.lastGranule;
    info[5] = (long)f.packets.size();
    return 0;
}

// The same walk with the entropy stage stopping at the SYMBOLS (CeltDecoder::decodeSymbols): records[max_frames] of
// nyqh_symbol_bytes_lm(channels, LM of the first frame) bytes each; the dump ends at the first frame of another size; info[6] = frames that carry host-built freq[]
// (NYQ_SYM_HOST_FREQ) instead of symbols.  A test hook: the GPU's band shapes against decode()'s.
long nyqh_symbol_bytes(int channels) { return (long)CeltDecoder::symbolBytes(channels); }
long nyqh_symbol_bytes_lm(int channels, int LM) { return (long)CeltDecoder::symbolBytes(channels, LM); }
// offsets16 != NULL: the records are written PACKED back to back (frame i at records + 16 * offsets16[i], max_frames + 1
// entries; `records` must hold max_frames slots all the same); NULL: one record per slot
int nyqh_decode_to_symbols_packed(const unsigned char *file, long size, long max_frames, unsigned char *records, unsigned *offsets16,
                                  int *flags, float *pf_gain, unsigned *range, long *info);
int nyqh_decode_to_symbols(const unsigned char *file, long size, long max_frames, unsigned char *records, int *flags, float *pf_gain,
                           unsigned *range, long *info) {
    return nyqh_decode_to_symbols_packed(file, size, max_frames, records, nullptr, flags, pf_gain, range, info);
}
int nyqh_decode_to_symbols_packed(const unsigned char *file, long size, long max_frames, unsigned char *records, unsigned *offsets16,
                                  int *flags, float *pf_gain, unsigned *range, long *info) {
    OggOpusFile f;
    try {
        f = parseOggOpus(file, (size_t)size);
    } catch (const std::exception &) {
        return -10;
    }
    if (f.head.mappingFamily != 0 || f.head.channels < 1 || f.head.channels > 2) return -11;
    const int CC = f.head.channels;
    size_t rec = 0;
    int frameSize = 0, LM = 0;
    CeltDecoder dec(CC);
    long nframes = 0, hostBuilt = 0;
    for (const auto &pkt : f.packets) {
        PacketFrames pf;
        if (!parseOpusPacket(pkt.data(), (int)pkt.size(), pf)) return -10;
        if (pf.config < 16) return -11;
        if (frameSize == 0) {
            frameSize = pf.frameSize;
            for (LM = 0; (120 << LM) != frameSize; LM++) {}
            rec = CeltDecoder::symbolBytes(CC, LM);
        }
        if (pf.frameSize != frameSize) goto done;        // a different frame size ends this fixed-shape dump
        dec.setEndBand(pf.bandwidthEnd);
        dec.setStreamChannels(pf.stereo ? 2 : 1);
        for (const auto &fr : pf.frames) {
            if (nframes >= max_frames) goto done;
            CeltFrame info1;
            if (offsets16 && nframes == 0) offsets16[0] = 0;
            unsigned char *r = offsets16 ? records + (size_t)offsets16[nframes] * 16 : records + (size_t)nframes * rec;
            const int rc = dec.decodeSymbols(fr.first, fr.second, pf.frameSize, r, info1);
            if (rc < 0) return rc;
            if (offsets16) offsets16[nframes + 1] = offsets16[nframes] + (unsigned)(info1.recordBytes / 16);
            hostBuilt += reinterpret_cast<const nyq_sym_head *>(r)->flags & NYQ_SYM_HOST_FREQ;
            flags[4 * nframes + 0] = info1.transient;
            flags[4 * nframes + 1] = info1.pfPitch;
            flags[4 * nframes + 2] = info1.pfTapset;
            flags[4 * nframes + 3] = info1.LM;
            pf_gain[nframes] = info1.pfGain;
            range[nframes] = info1.rangeFinal;
            nframes++;
        }
    }
done:
    info[0] = CC;
    info[1] = f.head.preSkip;
    info[2] = nframes;
    info[3] = frameSize;
    info[4] = (long)f.lastGranule;
    info[5] = (long)f.packets.size();
    info[6] = hostBuilt;
    info[7] = (long)rec;
    return 0;
}

// The device's own entropy stage (include/nyq_imdct.h nyq_celt_entropy_dev): its tables, and a file's frames as it takes them --
// every frame's bytes back to back in `payload` and a nyq_ent_desc per frame (one-stream files; the walk ends at the first frame
// of another size, as above).  info = {channels, pre-skip, frames, frame size, payload bytes}.  Returns 0, -10 (not Ogg Opus),
// -11 (not a one-stream CELT-only file), -12 (payload_cap too small).
long nyqh_device_entropy_frames(void) { return deviceEntropyFrames(); }
long nyqh_entropy_tables(void *out, long cap) {
    const long need = (long)entropyTablesBytes();
    if (!out) return need;
    if (cap < need) return -1;
    try {
        fillEntropyTables(out);
    } catch (const std::exception &e) {
        g_capi_err = e.what();
        return -1;
    }
    return need;
}
int nyqh_frame_table(const unsigned char *file, long size, long max_frames, unsigned char *payload, long payload_cap, nyq_ent_desc *desc, long *info) {
    OggOpusFile f;
    try {
        f = parseOggOpus(file, (size_t)size);
    } catch (const std::exception &) {
        return -10;
    }
    if (f.head.mappingFamily != 0 || f.head.channels < 1 || f.head.channels > 2) return -11;
    long nframes = 0, at = 0;
    int frameSize = 0;
    for (const auto &pkt : f.packets) {
        PacketFrames pf;
        if (!parseOpusPacket(pkt.data(), (int)pkt.size(), pf)) return -10;
        if (pf.config < 16) return -11;
        if (frameSize == 0) frameSize = pf.frameSize;
        if (pf.frameSize != frameSize) break;
        bool full = false;
        for (const auto &fr : pf.frames) {
            if (nframes >= max_frames) {
                full = true;
                break;
            }
            if (at + fr.second > payload_cap) return -12;
            if (fr.second > 0) std::memcpy(payload + at, fr.first, (size_t)fr.second);
            nyq_ent_desc &d = desc[nframes++];
            d.offset = (unsigned)at;
            d.len = (unsigned short)fr.second;
            d.channels = pf.stereo ? 2 : 1;
            d.start = 0;
            d.end = (unsigned char)pf.bandwidthEnd;
            d.pad[0] = d.pad[1] = d.pad[2] = 0;
            at += fr.second;
        }
        if (full) break;
    }
    info[0] = f.head.channels;
    info[1] = f.head.preSkip;
    info[2] = nframes;
    info[3] = frameSize;
    info[4] = at;
    return 0;
}

// NyquistIO::Load through the plugin surface (what examples/src/Main.cpp:86-154 does): returns the number
// of float samples (channels * frames) or <0 on exception; info = {channelCount, sampleRate, frameSize,
// lengthSeconds}.  Call twice: first with samples == NULL to learn the size.
long nyqh_nyquistio_load(const char *path, float *samples, long capacity, long *info) {
    try {
        nqr::NyquistIO loader;
        nqr::AudioData data;
        loader.Load(&data, std::string(path));
        info[0] = data.channelCount;
        info[1] = data.sampleRate;
        info[2] = (long)data.frameSize;
        info[3] = (long)data.lengthSeconds;
        if (samples && capacity >= (long)data.samples.size())
            std::memcpy(samples, data.samples.data(), data.samples.size() * sizeof(float));
        return (long)data.samples.size();
    } catch (const nqr::UnsupportedExtensionEx &) {
        return -2;
    } catch (const std::exception &e) {
        g_capi_err = e.what();
        return -1;
    }
}

// The same through NyquistIO::Load(AudioData*, const std::vector<uint8_t>&) (magic sniffing).
long nyqh_nyquistio_load_buffer(const unsigned char *file, long size, float *samples, long capacity, long *info) {
    try {
        nqr::NyquistIO loader;
        nqr::AudioData data;
        std::vector<uint8_t> buf(file, file + size);
        loader.Load(&data, buf);
        info[0] = data.channelCount;
        info[1] = data.sampleRate;
        info[2] = (long)data.frameSize;
        info[3] = (long)data.lengthSeconds;
        if (samples && capacity >= (long)data.samples.size())
            std::memcpy(samples, data.samples.data(), data.samples.size() * sizeof(float));
        return (long)data.samples.size();
    } catch (const nqr::UnsupportedExtensionEx &) {
        return -2;
    } catch (const std::exception &e) {
        g_capi_err = e.what();
        return -1;
    }
}

// The C entry points below share ONE decoder per device set, created on first use and kept for the life of the process
// (its page-locked staging memory and pooled sample buffers are reused; never destroyed, so no HIP call runs from a
// static destructor after the runtime has shut down).  Calls are serialised: decode() is not re-entrant.
static std::mutex g_capi_mu;
static nyq_host::BatchOpusDecoder *g_capi_dec = nullptr;
static std::vector<int> g_capi_devs;

// devices from NYQ_DEVICES ("0,1,2": one decoder over several GPUs) or NYQ_DEVICE (one), default device 0.  These entry
// points exist for the tests and tools, which set the variables before a call; anything that is not a list of
// non-negative integers is an error, not device 0.
static std::vector<int> parseDeviceList(const char *e) {
    std::vector<int> v;
    for (const char *p = e; *p;) {
        char *end = nullptr;
        const long d = std::strtol(p, &end, 10);
        if (end == p || d < 0 || d > 1023) throw std::runtime_error(std::string("bad device list: ") + e);
        v.push_back((int)d);
        p = end;
        if (*p == ',') { p++; if (!*p) throw std::runtime_error(std::string("bad device list: ") + e); }
        else if (*p) throw std::runtime_error(std::string("bad device list: ") + e);
    }
    if (v.empty()) throw std::runtime_error("empty device list");
    return v;
}
static std::vector<int> g_capi_devs_set;                       // nyqh_set_devices: overrides the environment (guarded by g_capi_mu)
static std::vector<int> capiDevices() {
    if (!g_capi_devs_set.empty()) return g_capi_devs_set;
    if (const char *e = std::getenv("NYQ_DEVICES")) return parseDeviceList(e);
    if (const char *e1 = std::getenv("NYQ_DEVICE")) return parseDeviceList(e1);
    return {0};
}

static nyq_host::BatchOpusDecoder &capiDecoder() {               // call with g_capi_mu held
    const std::vector<int> want = capiDevices();
    if (!g_capi_dec || want != g_capi_devs) {
        delete g_capi_dec;
        g_capi_dec = nullptr;
        g_capi_dec = new nyq_host::BatchOpusDecoder(want);
        g_capi_devs = want;
    }
    return *g_capi_dec;
}

// `count` copies of one file decoded as ONE batch (config 4 shape: many concurrent streams).  The results go through
// the sink form (every stream is looked at once and its buffer returns to the pool: what the reference's loop does
// with each AudioData), the first and the last decoded stream are copied out for checking.
// stats8 = {cpu_s, not_hidden_s, frames, threads, wall_s, devices, gpu_call_s, feeders}: wall_s is the whole call measured in
// here; gpu_call_s = time the feeder threads spent inside GPU calls (uploads, kernels, downloads), summed over the
// `feeders` feeder threads of all devices -- gpu_call_s / (wall_s * feeders) is how busy the GPU side was kept.
static long batch_decode_stats(const unsigned char *file, long size, long count, int threads, float *first, float *last,
                               long capacity, double *stats, int nstats) {
    try {
        const auto t0 = std::chrono::steady_clock::now();
        std::lock_guard<std::mutex> lk(g_capi_mu);
        std::vector<uint8_t> buf(file, file + size);
        std::vector<const std::vector<uint8_t> *> files((size_t)count, &buf);
        nyq_host::BatchOpusDecoder &dec = capiDecoder();
        nyq_host::BatchStats st;
        long nsamp = -1;
        bool failed = false;
        dec.decode(files, [&](size_t i, nyq_host::DecodedStream &d) {
            if (!d.error.empty()) { failed = true; return; }
            const long n = (long)d.pcm.size();
            if (i == 0) {
                nsamp = n;
                if (first && capacity >= n) std::memcpy(first, d.pcm.data(), (size_t)n * sizeof(float));
            }
            if (i + 1 == (size_t)count && last && capacity >= n) std::memcpy(last, d.pcm.data(), (size_t)n * sizeof(float));
        }, &st, threads);
        if (failed) return -1;
        stats[0] = st.cpuSeconds; stats[1] = st.gpuSeconds; stats[2] = (double)st.frames; stats[3] = st.threads;
        stats[4] = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
        stats[5] = (double)dec.deviceCount();
        if (nstats >= 8) {
            stats[6] = st.gpuBusySeconds;
            stats[7] = (double)(dec.deviceCount() * nyq_host::BatchOpusDecoder::kFeedersPerDevice);
        }
        return nsamp;
    } catch (const std::exception &e) {
        g_capi_err = e.what();
        return -1;
    }
}
long nyqh_batch_decode_timed(const unsigned char *file, long size, long count, int threads, float *first, float *last,
                             long capacity, double *stats6) {
    return batch_decode_stats(file, size, count, threads, first, last, capacity, stats6, 6);
}
long nyqh_batch_decode_stats8(const unsigned char *file, long size, long count, int threads, float *first, float *last,
                              long capacity, double *stats8) {
    return batch_decode_stats(file, size, count, threads, first, last, capacity, stats8, 8);
}

// the same with the original four-element stats = {cpu_s, not_hidden_s, frames, threads}
long nyqh_batch_decode(const unsigned char *file, long size, long count, int threads, float *first, float *last,
                       long capacity, double *stats) {
    double st6[6] = {0, 0, 0, 0, 0, 0};
    const long n = nyqh_batch_decode_timed(file, size, count, threads, first, last, capacity, st6);
    if (stats) std::memcpy(stats, st6, 4 * sizeof(double));
    return n;
}

// A batch of DIFFERENT files (mixed shapes: channel counts, frame sizes, lengths, multistream) as one call.
// nsamples[i] = interleaved sample count of file i (-1: that file failed), out = all decoded files back to back
// (if capacity suffices).  Returns the total sample count, or -1 if the call itself failed.
long nyqh_batch_decode_files(const unsigned char *const *files, const long *sizes, long count, int threads, long *nsamples,
                             float *out, long capacity) {
    try {
        std::lock_guard<std::mutex> lk(g_capi_mu);
        std::vector<std::vector<uint8_t>> bufs((size_t)count);
        std::vector<const std::vector<uint8_t> *> ptrs((size_t)count);
        for (long i = 0; i < count; i++) {
            bufs[i].assign(files[i], files[i] + sizes[i]);
            ptrs[i] = &bufs[i];
        }
        nyq_host::BatchOpusDecoder &dec = capiDecoder();
        std::vector<nyq_host::DecodedStream> res;
        dec.decode(ptrs, res, nullptr, threads);
        long total = 0;
        for (long i = 0; i < count; i++) {
            nsamples[i] = res[i].error.empty() ? (long)res[i].pcm.size() : -1;
            if (nsamples[i] > 0) total += nsamples[i];
        }
        if (out && capacity >= total) {
            long pos = 0;
            for (long i = 0; i < count; i++)
                if (nsamples[i] > 0) {
                    std::memcpy(out + pos, res[i].pcm.data(), (size_t)nsamples[i] * sizeof(float));
                    pos += nsamples[i];
                }
        }
        return total;
    } catch (const std::exception &e) {
        g_capi_err = e.what();
        return -1;
    }
}

// nqr::BatchLoad(out, buffers, devices) through the plugin surface: a batch of different files over a device LIST
// (a device may appear more than once: {0, 0} runs two device shards -- two sets of feeders, contexts and staging
// arenas -- on one GPU).  Outputs as nyqh_batch_decode_files; returns the total sample count, -1 if the call threw
// (one failing file makes BatchLoad throw after the others have been decoded, as documented in Decoders.h).
long nyqh_batch_load_devices(const unsigned char *const *files, const long *sizes, long count, const int *devices, int ndevices,
                             long *nsamples, float *out, long capacity) {
    try {
        std::vector<std::vector<uint8_t>> bufs((size_t)count);
        for (long i = 0; i < count; i++) bufs[i].assign(files[i], files[i] + sizes[i]);
        std::vector<nqr::AudioData> res;
        nqr::BatchLoad(res, bufs, std::vector<int>(devices, devices + ndevices));
        long total = 0;
        for (long i = 0; i < count; i++) {
            nsamples[i] = (long)res[i].samples.size();
            total += nsamples[i];
        }
        if (out && capacity >= total) {
            long pos = 0;
            for (long i = 0; i < count; i++) {
                std::memcpy(out + pos, res[i].samples.data(), (size_t)nsamples[i] * sizeof(float));
                pos += nsamples[i];
            }
        }
        return total;
    } catch (const std::exception &e) {
        g_capi_err = e.what();
        return -1;
    }
}

// device count of the decoder the batch entry points above are using right now (0: none made yet)
int nyqh_capi_device_count(void) {
    std::lock_guard<std::mutex> lk(g_capi_mu);
    return g_capi_dec ? g_capi_dec->deviceCount() : 0;
}

void nyqh_set_default_device(int device) { nqr::SetDefaultDevice(device); }

// device list of the batch entry points above from now on (instead of NYQ_DEVICES / NYQ_DEVICE); n = 0: back to the environment
void nyqh_set_devices(const int *devices, int n) {
    std::lock_guard<std::mutex> lk(g_capi_mu);
    g_capi_devs_set.assign(devices, devices + (n > 0 ? n : 0));
}

// counts[0] = decoders made, counts[1] = decoders torn down by the NyquistIO::Load / BatchLoad paths so far
void nyqh_decoder_pool_counts(long *counts) { nqr::DecoderPoolCounts(&counts[0], &counts[1]); }

// text of the last exception one of these entry points swallowed on the calling thread
const char *nyqh_last_error(void) { return g_capi_err.c_str(); }

}  // extern "C"
