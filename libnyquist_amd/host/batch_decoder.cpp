// batch_decoder.cpp -- see batch_decoder.hpp.
#include "batch_decoder.hpp"

#include <algorithm>
#include <atomic>
#include <chrono>
#include <cmath>
#include <cstring>
#include <map>
#include <stdexcept>
#include <thread>

#include "../../include/nyq_imdct.h"
#include "celt_decoder.hpp"
#include "opus_stream.hpp"

namespace nyq_host {

namespace {

// a run of consecutive frames of one size
struct Segment {
    int LM = 0;
    long nframes = 0;
    std::vector<float> freq;            // [nframes][channels][N]
    std::vector<uint8_t> transient;
    std::vector<int> pfPitch, pfTapset;
    std::vector<float> pfGain;
};

// what pass 1 leaves behind for one elementary (mono or coupled) Opus stream
struct StreamFrames {
    int channels = 0;
    std::vector<Segment> segs;          // usually one; encoders often close a stream with shorter frames
    long nframes = 0, transientCount = 0;
    int64_t samples = 0;                // decoded samples per channel over all segments
};

// one input file: its header and its elementary streams (one for mapping family 0)
struct FileJob {
    OpusHead head;
    int64_t lastGranule = -1;
    std::vector<StreamFrames> subs;
    std::string error;
};

void entropyDecode(const std::vector<uint8_t> &file, FileJob &job) {
    OggOpusFile f = parseOggOpus(file.data(), file.size());
    if (f.head.mappingFamily != 0 && f.head.mappingFamily != 1 && f.head.mappingFamily != 255)
        throw std::runtime_error("unsupported channel mapping family");
    job.head = f.head;
    job.lastGranule = f.lastGranule;
    const int S = f.head.streamCount;
    job.subs.assign(S, StreamFrames());
    std::vector<CeltDecoder> decs;
    decs.reserve(S);
    for (int k = 0; k < S; k++) {
        job.subs[k].channels = k < f.head.coupledCount ? 2 : 1;
        decs.emplace_back(job.subs[k].channels);
    }
    for (const auto &pkt : f.packets) {
        const uint8_t *p = pkt.data();
        int rem = (int)pkt.size();
        for (int k = 0; k < S; k++) {                       // opus_multistream_decoder.c:237-290
            PacketFrames pf;
            int used = rem;
            if (!parseOpusPacket(p, rem, pf, k != S - 1, &used)) throw std::runtime_error("malformed Opus packet");
            if (pf.config < 16) throw std::runtime_error("SILK/hybrid packet: only CELT-only streams are supported");
            StreamFrames &s = job.subs[k];
            int LM = 0;
            while ((120 << LM) != pf.frameSize) LM++;
            if (s.segs.empty() || s.segs.back().LM != LM) {
                s.segs.emplace_back();
                s.segs.back().LM = LM;
            }
            Segment &g = s.segs.back();
            decs[k].setEndBand(pf.bandwidthEnd);
            decs[k].setStreamChannels(pf.stereo ? 2 : 1);
            const size_t N = (size_t)pf.frameSize;
            for (const auto &fr : pf.frames) {
                g.freq.resize((size_t)(g.nframes + 1) * s.channels * N);
                CeltFrame info;
                const int rc = decs[k].decode(fr.first, fr.second, pf.frameSize, g.freq.data() + (size_t)g.nframes * s.channels * N, info);
                if (rc < 0) throw std::runtime_error("CELT frame failed to decode");
                g.transient.push_back(info.transient);
                g.pfPitch.push_back(info.pfPitch);
                g.pfTapset.push_back(info.pfTapset);
                g.pfGain.push_back(info.pfGain);
                s.transientCount += info.transient;
                g.nframes++;
                s.nframes++;
                s.samples += (int64_t)N;
            }
            p += used;
            rem -= used;
        }
    }
    for (const auto &s : job.subs)
        if (s.nframes == 0 || s.samples != job.subs[0].samples) throw std::runtime_error("no audio frames / streams of unequal length");
}

}  // namespace

BatchOpusDecoder::BatchOpusDecoder(int device) {
    nyq_ctx *c = nullptr;
    if (nyq_ctx_create(&c, device) != NYQ_OK)
        throw std::runtime_error(std::string("libnyq_imdct: ") + nyq_last_error(nullptr));
    ctx_ = c;
}

BatchOpusDecoder::~BatchOpusDecoder() { nyq_ctx_destroy((nyq_ctx *)ctx_); }

void BatchOpusDecoder::decode(const std::vector<const std::vector<uint8_t> *> &files, std::vector<DecodedStream> &out,
                              BatchStats *stats, int threads) {
    const size_t nfiles = files.size();
    out.assign(nfiles, DecodedStream());
    std::vector<FileJob> jobs(nfiles);
    if (threads <= 0) threads = (int)std::max(1u, std::thread::hardware_concurrency());
    threads = (int)std::min<size_t>(threads, std::max<size_t>(nfiles, 1));
    auto t0 = std::chrono::steady_clock::now();
    {   // pass 1: one file at a time per thread
        std::atomic<size_t> next{0};
        auto work = [&]() {
            for (size_t i = next++; i < nfiles; i = next++) {
                try {
                    entropyDecode(*files[i], jobs[i]);
                } catch (const std::exception &e) {
                    jobs[i].error = e.what();
                }
            }
        };
        std::vector<std::thread> pool;
        for (int t = 1; t < threads; t++) pool.emplace_back(work);
        work();
        for (auto &t : pool) t.join();
    }
    auto t1 = std::chrono::steady_clock::now();
    // the GPU batch is over ELEMENTARY streams: flatten (file, stream) pairs
    std::vector<StreamFrames *> sfp;
    std::vector<size_t> firstSub(nfiles, 0);
    for (size_t i = 0; i < nfiles; i++) {
        firstSub[i] = sfp.size();
        if (jobs[i].error.empty())
            for (auto &sub : jobs[i].subs) sfp.push_back(&sub);
    }
    const size_t n = sfp.size();
    struct Deref {                                         // keeps the code below reading sf[i].member
        std::vector<StreamFrames *> &v;
        StreamFrames &operator[](size_t i) { return *v[i]; }
    } sf{sfp};
    // pass 2: the first segment of every stream, grouped by (channels, LM) and padded to the longest,
    // goes to the GPU as one call per group; decoder state comes back so that the rare later segments
    // (a different frame size) continue exactly where the previous one stopped.
    nyq_ctx *ctx = (nyq_ctx *)ctx_;
    std::vector<std::vector<float>> pcmAll(n);            // interleaved, all segments, untrimmed
    std::vector<std::vector<float>> stateOf(n);           // per-stream decoder state (nstreams = 1 layout)
    std::map<std::pair<int, int>, std::vector<size_t>> groups;
    for (size_t i = 0; i < n; i++) groups[{sf[i].channels, sf[i].segs[0].LM}].push_back(i);
    long totalFrames = 0;
    for (auto &g : groups) {
        const int ch = g.first.first, LM = g.first.second;
        const size_t N = (size_t)120 << LM;
        const std::vector<size_t> &ids = g.second;
        size_t maxF = 0;
        for (size_t i : ids) maxF = std::max(maxF, (size_t)sf[i].segs[0].nframes);
        const size_t ns = ids.size(), nsc = ns * ch;
        std::vector<float> freq(ns * maxF * ch * N, 0.f), pcm(ns * maxF * N * ch);
        std::vector<uint8_t> tr(ns * maxF, 0);
        std::vector<int> pp(ns * maxF, 0), pt(ns * maxF, 0);
        std::vector<float> pg(ns * maxF, 0.f);
        bool anyMore = false;
        for (size_t k = 0; k < ns; k++) {
            const Segment &s = sf[ids[k]].segs[0];
            std::memcpy(&freq[k * maxF * ch * N], s.freq.data(), s.freq.size() * sizeof(float));
            std::memcpy(&tr[k * maxF], s.transient.data(), s.transient.size());
            std::memcpy(&pp[k * maxF], s.pfPitch.data(), s.pfPitch.size() * sizeof(int));
            std::memcpy(&pt[k * maxF], s.pfTapset.data(), s.pfTapset.size() * sizeof(int));
            std::memcpy(&pg[k * maxF], s.pfGain.data(), s.pfGain.size() * sizeof(float));
            totalFrames += s.nframes;
            anyMore |= sf[ids[k]].segs.size() > 1;
        }
        // the state is only meaningful for streams whose first segment fills the whole padded length;
        // streams that continue with another segment are therefore given their own call when they are shorter
        std::vector<float> state(anyMore ? nyq_celt_state_floats(ns, ch) : 0, 0.f);
        if (nyq_celt_frames_to_pcm(ctx, LM, freq.data(), tr.data(), pp.data(), pg.data(), pt.data(), pcm.data(),
                                   anyMore ? state.data() : nullptr, ns, maxF, ch) != NYQ_OK)
            throw std::runtime_error(std::string("libnyq_imdct: ") + nyq_last_error(ctx));
        for (size_t k = 0; k < ns; k++) {
            const size_t i = ids[k];
            const Segment &s = sf[i].segs[0];
            pcmAll[i].assign(&pcm[k * maxF * N * ch], &pcm[k * maxF * N * ch] + (size_t)s.nframes * N * ch);
            if (sf[i].segs.size() > 1) {
                if ((size_t)s.nframes != maxF) {           // padded with silent frames: redo alone for an exact state
                    std::vector<float> st1(nyq_celt_state_floats(1, ch), 0.f), out1((size_t)s.nframes * N * ch);
                    if (nyq_celt_frames_to_pcm(ctx, LM, s.freq.data(), s.transient.data(), s.pfPitch.data(), s.pfGain.data(),
                                               s.pfTapset.data(), out1.data(), st1.data(), 1, (size_t)s.nframes, ch) != NYQ_OK)
                        throw std::runtime_error(std::string("libnyq_imdct: ") + nyq_last_error(ctx));
                    stateOf[i] = st1;
                } else {                                   // slice stream k out of the group state
                    std::vector<float> st1(nyq_celt_state_floats(1, ch));
                    const float *ov = state.data(), *hi = ov + nsc * 60, *de = hi + nsc * 1088, *pf = de + nsc;
                    float *o = st1.data();
                    std::memcpy(o, ov + k * ch * 60, sizeof(float) * ch * 60); o += ch * 60;
                    std::memcpy(o, hi + k * ch * 1088, sizeof(float) * ch * 1088); o += ch * 1088;
                    std::memcpy(o, de + k * ch, sizeof(float) * ch); o += ch;
                    std::memcpy(o, pf + k * 6, sizeof(float) * 6);
                    stateOf[i] = st1;
                }
            }
        }
    }
    for (size_t i = 0; i < n; i++) {                       // later segments, one stream at a time
        const int ch = sf[i].channels;
        for (size_t g = 1; g < sf[i].segs.size(); g++) {
            const Segment &s = sf[i].segs[g];
            const size_t N = (size_t)120 << s.LM;
            std::vector<float> out1((size_t)s.nframes * N * ch);
            if (nyq_celt_frames_to_pcm(ctx, s.LM, s.freq.data(), s.transient.data(), s.pfPitch.data(), s.pfGain.data(),
                                       s.pfTapset.data(), out1.data(), stateOf[i].data(), 1, (size_t)s.nframes, ch) != NYQ_OK)
                throw std::runtime_error(std::string("libnyq_imdct: ") + nyq_last_error(ctx));
            pcmAll[i].insert(pcmAll[i].end(), out1.begin(), out1.end());
            totalFrames += s.nframes;
        }
    }
    // pass 3: channel mapping (opus_multistream_decoder.c:305-331), then trimming (opusfile: skip pre_skip
    // samples, stop at the last page's granule position) and the header gain
    for (size_t i = 0; i < nfiles; i++) {
        if (!jobs[i].error.empty()) continue;
        const FileJob &job = jobs[i];
        DecodedStream &d = out[i];
        const int ch = job.head.channels;
        d.channels = ch;
        d.preSkip = job.head.preSkip;
        const int64_t decoded = job.subs[0].samples;
        for (const auto &sub : job.subs) {
            d.frames += sub.nframes;
            d.transientFrames += sub.transientCount;
        }
        const int64_t endSample = job.lastGranule >= 0 ? std::min<int64_t>(decoded, job.lastGranule) : decoded;
        int64_t total = endSample - job.head.preSkip;
        if (total < 0) total = 0;
        d.totalSamples = total;
        d.pcm.assign((size_t)total * ch, 0.f);
        const float gain = job.head.outputGainQ8 == 0 ? 1.f   // OPUS_SET_GAIN, opus_decoder_clean.c:700-712
                                                       : (float)std::exp(0.6931471805599453094 * (6.48814081e-4 * job.head.outputGainQ8));
        for (int c = 0; c < ch; c++) {
            const int idx = job.head.mapping[c];
            if (idx == 255) continue;                      // silent channel
            int sub, sc;
            if (idx < 2 * job.head.coupledCount) { sub = idx / 2; sc = idx & 1; }
            else { sub = idx - job.head.coupledCount; sc = 0; }
            const size_t flat = firstSub[i] + (size_t)sub;
            const int sch = sf[flat].channels;
            const float *src = pcmAll[flat].data() + (size_t)job.head.preSkip * sch + sc;
            float *dst = d.pcm.data() + c;
            if (gain == 1.f)
                for (int64_t t = 0; t < total; t++) dst[t * ch] = src[t * sch];
            else
                for (int64_t t = 0; t < total; t++) dst[t * ch] = src[t * sch] * gain;
        }
    }
    auto t2 = std::chrono::steady_clock::now();
    for (size_t i = 0; i < nfiles; i++)
        if (!jobs[i].error.empty()) out[i].error = jobs[i].error;
    if (stats) {
        stats->cpuSeconds = std::chrono::duration<double>(t1 - t0).count();
        stats->gpuSeconds = std::chrono::duration<double>(t2 - t1).count();
        stats->frames = totalFrames;
        stats->threads = threads;
    }
}

}  // namespace nyq_host
