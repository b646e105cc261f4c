// batch_decoder.cpp -- see batch_decoder.hpp.
#include "batch_decoder.hpp"

#include <algorithm>
#include <atomic>
#include <chrono>
#include <cstdio>
#include <cmath>
#include <condition_variable>
#include <cstdlib>
#include <cstring>
#include <deque>
#include <functional>
#include <map>
#include <mutex>
#include <exception>
#include <stdexcept>
#include <system_error>
#include <thread>
#include <tuple>

#include <sched.h>

#include "../../include/nyq_imdct.h"
#include "celt_decoder.hpp"
#include "opus_stream.hpp"

namespace nyq_host {

// Test-only fault injection (tests/sched/lease_check.cpp, built with -DNYQ_HOST_TEST_HOOKS): the n-th thread start
// from now on fails the way pthread_create does when the process is out of threads.  Never compiled into the product.
#ifdef NYQ_HOST_TEST_HOOKS
std::atomic<long> g_failThreadStartIn{0};
#endif

namespace {

// a run of consecutive frames of one size
struct SegPlan {
    int LM = 0;
    long nframes = 0;
};

// decoded storage of a segment that is NOT the first of its stream (rare: encoders often close a stream
// with shorter frames); first segments are decoded in place into the group buffers below
struct Segment {
    int LM = 0;
    long nframes = 0;
    std::vector<float> freq;            // [nframes][channels][N]
    std::vector<uint8_t> transient;
    std::vector<int> pfPitch, pfTapset;
    std::vector<float> pfGain;
    std::vector<uint8_t> bytes;         // (device entropy stage: the frames' bytes in slots of nyq_celt_byte_slot, and a word per frame)
    std::vector<uint32_t> words;
};

// one elementary (mono or coupled) Opus stream
struct StreamFrames {
    int channels = 0;
    std::vector<SegPlan> plan;          // from the scan: frame sizes and counts, nothing decoded yet
    std::vector<Segment> later;         // plan[1..] decoded
    // where the first segment goes (set by the layout step): slot `slot` of group `group`
    size_t group = 0, slot = 0, piece = 0;
    // input of the GPU stages, one record per frame: freq[channels][N] floats, or -- symbols -- the frame's symbol record
    // (nyq_celt_symbol_bytes: 20 ms mono / stereo streams, whose band shapes the GPU builds itself)
    uint8_t *in0 = nullptr;
    size_t frameBytes = 0;
    bool symbols = false;
    uint32_t *off0 = nullptr;           // symbols: frame i's record at in0 + 16 * off0[i] (off0[i + 1] = its end)
    bool packed = true;                 // ... back to back (a record is as long as its content), or one per slot of frameBytes
    bool bytesMode = false;             // the GPU runs the entropy stage too: in0 holds the frames' bytes (slots of frameBytes), pp0 a word per frame
    uint8_t *tr0 = nullptr;
    int *pp0 = nullptr, *pt0 = nullptr;
    float *pg0 = nullptr;
    long nframes = 0, transientCount = 0;
    int64_t samples = 0;                // decoded samples per channel over all segments
    // progress of pass 1 on the first segment, published every `sliceLen` frames so that the GPU can start on
    // time slices of long streams while the rest is still being entropy-decoded
    std::atomic<long> *progress = nullptr;
    long sliceLen = 0;
};

std::atomic<long> g_deviceEntropyFrames{0};   // frames (padding included) whose entropy stage ran on the GPU, process-wide (a test hook)

// one input file: its demultiplexed packets, header and elementary streams (one for mapping family 0)
struct FileJob {
    OggOpusFile f;
    std::vector<StreamFrames> subs;
    std::string error;
};

// pass 0: container and packet framing only -- how many frames of which size every stream holds
void scanFile(const std::vector<uint8_t> &file, FileJob &job) {
    job.f = parseOggOpus(file.data(), file.size());
    const OpusHead &h = job.f.head;
    if (h.mappingFamily != 0 && h.mappingFamily != 1 && h.mappingFamily != 255)
        throw std::runtime_error("unsupported channel mapping family");
    const int S = h.streamCount;
    job.subs.assign(S, StreamFrames());
    for (int k = 0; k < S; k++) job.subs[k].channels = k < h.coupledCount ? 2 : 1;
    PacketFrames pf;                                        // (one frame list for the whole file: no allocation per packet)
    for (const auto &pkt : job.f.packets) {
        const uint8_t *p = pkt.data();
        int rem = (int)pkt.size();
        for (int k = 0; k < S; k++) {                       // opus_multistream_decoder.c:237-290
            int used = rem;
            if (!parseOpusPacket(p, rem, pf, k != S - 1, &used)) throw std::runtime_error("malformed Opus packet");
            if (pf.config < 16) throw std::runtime_error("SILK/hybrid packet: only CELT-only streams are supported");
            StreamFrames &s = job.subs[k];
            int LM = 0;
            while ((120 << LM) != pf.frameSize) LM++;
            if (s.plan.empty() || s.plan.back().LM != LM) s.plan.push_back({LM, 0});
            s.plan.back().nframes += (long)pf.frames.size();
            s.nframes += (long)pf.frames.size();
            s.samples += (int64_t)pf.frames.size() * pf.frameSize;
            p += used;
            rem -= used;
        }
    }
    for (const auto &s : job.subs)
        if (s.nframes == 0 || s.samples != job.subs[0].samples) throw std::runtime_error("no audio frames / streams of unequal length");
}

// pass 1: the bit-serial half of every frame (range decoder ... denormalisation), first segments straight
// into the group buffers the GPU reads
template <class OnSlice>
void entropyDecode(FileJob &job, OnSlice &&onSlice) {
    const int S = job.f.head.streamCount;
    std::vector<CeltDecoder> decs;
    decs.reserve(S);
    std::vector<size_t> seg(S, 0);
    std::vector<long> done(S, 0);                           // frames decoded in the current segment
    for (int k = 0; k < S; k++) {
        StreamFrames &s = job.subs[k];
        decs.emplace_back(s.channels);
        s.later.resize(s.plan.size() - 1);
        for (size_t g = 1; g < s.plan.size(); g++) {
            Segment &L = s.later[g - 1];
            L.LM = s.plan[g].LM;
            L.nframes = s.plan[g].nframes;
            if (s.bytesMode) {
                L.bytes.assign((size_t)L.nframes * nyq_celt_byte_slot(), 0);
                L.words.assign((size_t)L.nframes, 0);
                continue;
            }
            L.freq.resize((size_t)L.nframes * s.channels * ((size_t)120 << L.LM));
            L.transient.resize(L.nframes);
            L.pfPitch.resize(L.nframes);
            L.pfTapset.resize(L.nframes);
            L.pfGain.resize(L.nframes);
        }
    }
    PacketFrames pf;
    for (const auto &pkt : job.f.packets) {
        const uint8_t *p = pkt.data();
        int rem = (int)pkt.size();
        for (int k = 0; k < S; k++) {
            int used = rem;
            parseOpusPacket(p, rem, pf, k != S - 1, &used);  // validated by the scan
            StreamFrames &s = job.subs[k];
            decs[k].setEndBand(pf.bandwidthEnd);
            decs[k].setStreamChannels(pf.stereo ? 2 : 1);
            const size_t N = (size_t)pf.frameSize;
            for (const auto &fr : pf.frames) {
                if (done[k] == s.plan[seg[k]].nframes) { seg[k]++; done[k] = 0; }
                const long i = done[k]++;
                Segment *L = seg[k] ? &s.later[seg[k] - 1] : nullptr;
                CeltFrame info;
                int rc;
                if (s.bytesMode) {                                   // the packet walk is all the host does: the frame's bytes and what its packet says
                    const unsigned word = (unsigned)fr.second | (unsigned)(pf.stereo ? 2 : 1) << 16 | (unsigned)pf.bandwidthEnd << 24;
                    if (L) {
                        if (fr.second > 0) std::memcpy(L->bytes.data() + (size_t)i * nyq_celt_byte_slot(), fr.first, (size_t)fr.second);
                        L->words[(size_t)i] = word;
                        continue;
                    }
                    if (fr.second > 0) std::memcpy(s.in0 + (size_t)i * s.frameBytes, fr.first, (size_t)fr.second);
                    s.pp0[i] = (int)word;
                    if (s.sliceLen > 0 && (i + 1) % s.sliceLen == 0) {
                        s.progress->store(i + 1, std::memory_order_release);
                        onSlice(s);
                    }
                    continue;
                }
                if (!L && s.symbols) {
                    rc = decs[k].decodeSymbols(fr.first, fr.second, pf.frameSize, s.in0 + (size_t)s.off0[i] * 16, info);
                    s.off0[i + 1] = s.off0[i] + (uint32_t)((s.packed ? info.recordBytes : s.frameBytes) / 16);
                } else {
                    float *dst = L ? L->freq.data() + (size_t)i * s.channels * N : reinterpret_cast<float *>(s.in0 + (size_t)i * s.frameBytes);
                    rc = decs[k].decode(fr.first, fr.second, pf.frameSize, dst, info);
                }
                if (rc < 0) throw std::runtime_error("CELT frame failed to decode");
                (L ? L->transient.data() : s.tr0)[i] = info.transient;
                (L ? L->pfPitch.data() : s.pp0)[i] = info.pfPitch;
                (L ? L->pfTapset.data() : s.pt0)[i] = info.pfTapset;
                (L ? L->pfGain.data() : s.pg0)[i] = info.pfGain;
                s.transientCount += info.transient;
                if (!L && s.sliceLen > 0 && (i + 1) % s.sliceLen == 0) {   // a whole time slice of this stream is ready
                    s.progress->store(i + 1, std::memory_order_release);
                    onSlice(s);
                }
            }
            p += used;
            rem -= used;
        }
    }
}

// streams of one shape (channels, frame size of the first segment), padded to the longest
struct Group {
    bool mapped = false;                // its streams write through output records into their files' device buffers
    int ch = 0, LM = 0, dev = 0;        // dev: index into the decoder's device list
    bool symbols = false;               // the entropy stage stops at the symbols: the GPU builds the band shapes (20 ms, <= 2 channels)
    bool bytes = false;                 // the entropy stage itself runs on the GPU: `in` holds the frames' bytes, `pp` their words
    size_t N = 0, ns = 0, maxF = 0;
    size_t frameBytes = 0;              // of GPU input per frame: freq[ch][N] or a symbol record
    std::vector<size_t> ids;            // flattened stream indices, slot order
    uint8_t *in = nullptr;
    uint32_t *off = nullptr;            // symbols: [ns][maxF + 1] record offsets (16-byte units) inside each stream's maxF * frameBytes region
    float *out = nullptr, *pg = nullptr;
    int *pp = nullptr, *pt = nullptr;
    uint8_t *tr = nullptr;
};

// a run of slots of one group, walked by the GPU in time slices of `sliceLen` frames: slice s goes out as soon as
// every stream of the piece has been decoded that far (decoder state carried from slice to slice on the host)
struct Piece {
    size_t group = 0, k0 = 0, k1 = 0;
    size_t sliceLen = 0, nslices = 1, nextSlice = 0;   // nextSlice / inFlight / appendTurn: guarded by the scheduler's mutex
    size_t appendTurn = 0;              // slices hand their samples to the files in order
    bool inFlight = false;
    bool anyMore = false;               // some stream continues with another segment: keep the decoder state
    std::vector<float> state;           // nyq_celt_state_floats(k1 - k0, channels), zero = fresh decoder
};

constexpr size_t kPieceBytes = (size_t)24 << 20;   // of freq per piece (short streams)
constexpr size_t kSliceBytes = (size_t)32 << 20;   // of freq per time slice of a piece (long streams)

size_t align256(size_t v) { return (v + 255) & ~(size_t)255; }

// Host threads this process may really use: the affinity mask, clipped by the cgroup CPU quota (a container
// that sees 256 CPUs but is allowed 16 must not start 256 decoding threads).
int usableHostThreads() {
    int n = 0;
    cpu_set_t set;
    if (sched_getaffinity(0, sizeof set, &set) == 0) n = CPU_COUNT(&set);
    if (n <= 0) n = (int)std::max(1u, std::thread::hardware_concurrency());
    if (FILE *f = std::fopen("/sys/fs/cgroup/cpu.max", "r")) {            // cgroup v2: "<quota|max> <period>"
        char quota[32];
        long period = 0;
        if (std::fscanf(f, "%31s %ld", quota, &period) == 2 && period > 0 && std::strcmp(quota, "max") != 0)
            n = std::min(n, std::max(1, (int)((std::atol(quota) + period / 2) / period)));
        std::fclose(f);
    } else {                                                              // cgroup v1
        long q = -1, p = 0;
        if (FILE *fq = std::fopen("/sys/fs/cgroup/cpu/cpu.cfs_quota_us", "r")) { if (std::fscanf(fq, "%ld", &q) != 1) q = -1; std::fclose(fq); }
        if (FILE *fp = std::fopen("/sys/fs/cgroup/cpu/cpu.cfs_period_us", "r")) { if (std::fscanf(fp, "%ld", &p) != 1) p = 0; std::fclose(fp); }
        if (q > 0 && p > 0) n = std::min(n, std::max(1, (int)((q + p / 2) / p)));
    }
    return n;
}

// Threads that are ALWAYS joined: a std::thread that is still joinable when it is destroyed ends the process
// (std::terminate -> abort), so nothing here may leave a scope -- by return or by exception -- with one running.
class JoinedThreads {
public:
    JoinedThreads() = default;
    JoinedThreads(const JoinedThreads &) = delete;
    JoinedThreads &operator=(const JoinedThreads &) = delete;
    ~JoinedThreads() { join(); }
    template <class Fn>
    void start(Fn &&fn) {                   // throws std::system_error like std::thread; what has been started stays owned
#ifdef NYQ_HOST_TEST_HOOKS
        if (g_failThreadStartIn.load() > 0 && g_failThreadStartIn.fetch_sub(1) == 1)
            throw std::system_error(std::make_error_code(std::errc::resource_unavailable_try_again), "thread start (injected)");
#endif
        threads_.emplace_back(std::forward<Fn>(fn));
    }
    void join() {
        for (auto &t : threads_)
            if (t.joinable()) t.join();
        threads_.clear();
    }
    size_t size() const { return threads_.size(); }

private:
    std::vector<std::thread> threads_;
};

// body(i) for i in [0, n) on up to `threads` threads (the caller's included).  Nothing escapes a worker thread: the first
// exception of any body is kept, the remaining indices are dropped, every thread is joined, and it is rethrown here.  If
// the process cannot start another thread the loop runs on the threads it got (the calling thread at least).
template <class F>
void parallelFor(size_t n, int threads, F &&body) {
    std::atomic<size_t> next{0};
    std::mutex emu;
    std::exception_ptr first;
    auto work = [&]() {
        for (size_t i = next++; i < n; i = next++) {
            try {
                body(i);
            } catch (...) {
                std::lock_guard<std::mutex> lk(emu);
                if (!first) first = std::current_exception();
                next.store(n);
            }
        }
    };
    {
        JoinedThreads pool;
        const int extra = (int)std::min<size_t>(threads > 0 ? threads - 1 : 0, n > 0 ? n - 1 : 0);
        try {
            for (int t = 0; t < extra; t++) pool.start(work);
        } catch (const std::system_error &) {
            // fewer threads than asked for: slower, not wrong
        }
        work();
    }
    if (first) std::rethrow_exception(first);
}


// copy the decoder state of stream `k` out of / into the layout nyq_celt_state_floats(ns, ch) describes:
// [overlap ns*ch*60 | post-filter history ns*ch*1088 | de-emphasis ns*ch | post-filter parameters ns*6]
void stateOfStream(const float *batch, size_t ns, int ch, size_t k, float *one) {
    const size_t nsc = ns * ch;
    const float *ov = batch, *hi = ov + nsc * 60, *de = hi + nsc * 1088, *pf = de + nsc;
    std::memcpy(one, ov + k * ch * 60, sizeof(float) * ch * 60); one += ch * 60;
    std::memcpy(one, hi + k * ch * 1088, sizeof(float) * ch * 1088); one += ch * 1088;
    std::memcpy(one, de + k * ch, sizeof(float) * ch); one += ch;
    std::memcpy(one, pf + k * 6, sizeof(float) * 6); one += 6;
    const size_t es = nyq_celt_state_floats(1, ch) - ((size_t)ch * (60 + 1088 + 1) + 6);   // (the entropy stage's state, where the library has one)
    std::memcpy(one, pf + ns * 6 + k * es, sizeof(float) * es);
}
void stateIntoBatch(float *batch, size_t ns, int ch, size_t k, const float *one) {
    const size_t nsc = ns * ch;
    float *ov = batch, *hi = ov + nsc * 60, *de = hi + nsc * 1088, *pf = de + nsc;
    std::memcpy(ov + k * ch * 60, one, sizeof(float) * ch * 60); one += ch * 60;
    std::memcpy(hi + k * ch * 1088, one, sizeof(float) * ch * 1088); one += ch * 1088;
    std::memcpy(de + k * ch, one, sizeof(float) * ch); one += ch;
    std::memcpy(pf + k * 6, one, sizeof(float) * 6); one += 6;
    const size_t es = nyq_celt_state_floats(1, ch) - ((size_t)ch * (60 + 1088 + 1) + 6);
    std::memcpy(pf + ns * 6 + k * es, one, sizeof(float) * es);
}

// One memory-bounded sub-batch of files on its way through passes 1-3 (see batch_decoder.hpp).
class SubBatch {
public:
    // ctxs: ndev * feedersPerDev contexts, device d's at [d * feedersPerDev, ...); arena(dev, bytes) = that device's staging memory
    SubBatch(std::vector<FileJob> &jobs, std::vector<DecodedStream> &out, const std::vector<size_t> &members, void *const *ctxs,
             int ndev, int feedersPerDev, int threads, const std::function<void *(int, size_t)> &arena,
             const std::function<void *(int, size_t)> &devArena)
        : jobs_(jobs), out_(out), members_(members), ctxs_(ctxs), ndev_(ndev), feedersPerDev_(feedersPerDev),
          nfeeders_(ndev * feedersPerDev), threads_(threads), arena_(arena), devArena_(devArena), firstSub_(jobs.size(), 0),
          finished_(jobs.size(), 0), streamed_(jobs.size(), 0), window_(jobs.size()), subsLeft_(jobs.size()), fileDev_(jobs.size(), 0),
          devOut_(jobs.size(), nullptr), hostOut_(jobs.size(), nullptr), ready_((size_t)ndev), gpuBusy_((size_t)(ndev * feedersPerDev), 0.0) {}

    // 20 ms mono / stereo streams hand SYMBOL records to the GPU (their band shapes are built there); off: freq[] as for every other shape
    bool symbolRecords_ = true;
    bool deviceEntropy_ = false;        // NYQ_DEVICE_ENTROPY: eligible streams hand the GPU their frames' bytes
    bool symbolRecords() const { return symbolRecords_; }
    bool packedRecords_ = false;        // records back to back (half the upload) or one per slot (one strided copy per window)
    size_t pieceBytes_ = kPieceBytes;
    size_t longPieceStreams_ = 0;
    bool trace_ = false;

    // returns when every file of the sub-batch is decoded (or has its error set); throws if the GPU failed
    void run() {
        const auto tb = std::chrono::steady_clock::now();
        flatten();
        planOutput();
        layout();
        placeOutput();
        std::chrono::steady_clock::time_point t1, tDecode0;
        {
            // The feeders are joined whichever way this block is left.  If anything throws in here (a feeder that cannot
            // be started, an exception out of a decoding thread) the guard below raises `abort_` first, so that feeders
            // waiting for work return instead of waiting for pieces that will never come.
            JoinedThreads feeders;
            struct AbortOnUnwind {
                SubBatch &sb;
                bool armed = true;
                ~AbortOnUnwind() {
                    if (!armed) return;
                    {
                        std::lock_guard<std::mutex> lk(sb.mu_);
                        sb.abort_ = true;
                    }
                    sb.cv_.notify_all();
                }
            } guard{*this};
            // one feeder per device first, then the second of each, ...: if the process runs out of threads on the way
            // the batch goes on with the feeders it has, as long as every device has one
            try {
                for (int f = 0; f < feedersPerDev_; f++)
                    for (int d = 0; d < ndev_; d++) {
                        const int which = d * feedersPerDev_ + f;
                        feeders.start([this, which] { feederLoop(which); });
                    }
            } catch (const std::system_error &e) {
                if (feeders.size() < (size_t)ndev_) throw std::runtime_error(std::string("cannot start a feeder thread per device: ") + e.what());
            }
            tDecode0 = std::chrono::steady_clock::now();
            parallelFor(members_.size(), threads_, [&](size_t mi) { decodeFile(members_[mi]); });
            decodeDone_.store(true, std::memory_order_release);
            t1 = std::chrono::steady_clock::now();
            guard.armed = false;
            cv_.notify_all();
        }
        const auto tJoined = std::chrono::steady_clock::now();
        if (!gpuError_.empty()) throw std::runtime_error(gpuError_);
        laterSegments();
        const auto tLater = std::chrono::steady_clock::now();
        // pass 3 for the files that could not be finished as their pieces completed (later segments, or a stream of
        // the file in a piece that ended after the file's other streams)
        parallelFor(members_.size(), threads_, [&](size_t mi) {
            const size_t i = members_[mi];
            // (the feeders are gone: context k of the file's device, by worker index, under that context's mutex)
            if (jobs_[i].error.empty() && !finished_[i]) finishFile(i, fileDev_[i] * feedersPerDev_ + (int)(mi % (size_t)feedersPerDev_));
        });
        cpuSeconds = std::chrono::duration<double>(t1 - tb).count();
        tailSeconds = std::chrono::duration<double>(std::chrono::steady_clock::now() - t1).count();
        if (trace_) {                                           // NYQ_BATCH_TRACE=1: where a sub-batch's time goes (stderr)
            auto ms = [](std::chrono::steady_clock::time_point a, std::chrono::steady_clock::time_point b) {
                return std::chrono::duration<double, std::milli>(b - a).count();
            };
            std::fprintf(stderr, "[nyq batch] %zu files, %zu pieces: set-up %.1f ms, entropy stage %.1f, feeders drained +%.1f, later segments +%.1f, "
                                 "finish +%.1f\n", members_.size(), pieces_.size(), ms(tb, tDecode0), ms(tDecode0, t1), ms(t1, tJoined),
                         ms(tJoined, tLater), ms(tLater, std::chrono::steady_clock::now()));
        }
        for (double b : gpuBusy_) busySeconds += b;
    }

    double cpuSeconds = 0, tailSeconds = 0, busySeconds = 0;
    std::atomic<long> frames{0};

private:
    StreamFrames &sf(size_t flat) { return *sfp_[flat]; }

    // the GPU batch is over ELEMENTARY streams: flatten (file, stream) pairs
    void flatten() {
        for (size_t i : members_) {
            firstSub_[i] = sfp_.size();
            if (jobs_[i].error.empty())
                for (auto &sub : jobs_[i].subs) sfp_.push_back(&sub);
        }
        fileOf_.resize(sfp_.size());
        for (size_t i : members_) {
            subsLeft_[i].store(jobs_[i].error.empty() ? (int)jobs_[i].subs.size() : 0, std::memory_order_relaxed);
            if (jobs_[i].error.empty())
                for (size_t k = 0; k < jobs_[i].subs.size(); k++) fileOf_[firstSub_[i] + k] = i;
        }
        stateOf_.resize(sfp_.size());
        laterPcm_.resize(sfp_.size());
    }

    // layout: the first segment of every stream, grouped by (channels, LM) and padded to the longest; a group is cut
    // into pieces of consecutive slots, a piece into time slices
    void layout() {
        const size_t n = sfp_.size();
        // Elementary stream number s of a (channels, frame size) class goes to device s mod G:
        // a group -- one set of staging buffers, one shape of GPU calls -- belongs to ONE device.
        // The streams of a file whose samples are mapped on the device (several streams, a permuted mapping, a header gain)
        // all go to ONE device -- the file's interleaved buffer lives there -- and form groups of their own.
        std::map<std::pair<int, int>, size_t> seenOfClass;
        std::map<std::tuple<int, int, int, int>, size_t> groupOf;
        for (size_t i = 0; i < n; i++) {
            const bool mapped = !streamed_[fileOf_[i]];
            const int dev = mapped ? fileDev_[fileOf_[i]] : (int)(seenOfClass[{sf(i).channels, sf(i).plan[0].LM}]++ % (size_t)ndev_);
            const bool byBytes = deviceEntropy_ && sf(i).channels <= 2;
            const std::tuple<int, int, int, int> key{sf(i).channels, sf(i).plan[0].LM, dev, (mapped ? 1 : 0) | (byBytes ? 2 : 0)};
            auto it = groupOf.find(key);
            if (it == groupOf.end()) {
                it = groupOf.emplace(key, groups_.size()).first;
                groups_.emplace_back();
                groups_.back().ch = std::get<0>(key);
                groups_.back().LM = std::get<1>(key);
                groups_.back().N = (size_t)120 << std::get<1>(key);
                groups_.back().dev = dev;
                groups_.back().mapped = mapped;
                groups_.back().bytes = byBytes;
                groups_.back().symbols = !byBytes && symbolRecords() && groups_.back().ch <= 2;
                groups_.back().frameBytes = byBytes ? nyq_celt_byte_slot()
                                            : groups_.back().symbols ? nyq_celt_symbol_bytes_lm(groups_.back().ch, groups_.back().LM)
                                                                     : (size_t)groups_.back().ch * groups_.back().N * sizeof(float);
            }
            Group &g = groups_[it->second];
            sf(i).group = it->second;
            sf(i).slot = g.ids.size();
            g.ids.push_back(i);
            g.maxF = std::max(g.maxF, (size_t)sf(i).plan[0].nframes);
        }
        std::vector<size_t> bytes((size_t)ndev_, 0);
        for (Group &g : groups_) {
            g.ns = g.ids.size();
            const size_t x = g.ns * g.maxF * g.ch * g.N * sizeof(float), q = g.ns * g.maxF;
            bytes[(size_t)g.dev] += align256(q * g.frameBytes) + (g.mapped ? 0 : 1) * align256(x) + 3 * align256(q * 4) + align256(q) +   // (mapped: no dense output)
                                    (g.symbols ? align256(g.ns * (g.maxF + 1) * 4) : 0);
        }
        // a mapped file's samples come back from its device buffer through page-locked memory too (one DMA per file; a copy
        // into ordinary memory goes through the runtime's own staging in chunks, a blit kernel each: 128 surround files
        // spent 10 ms of the job's 20 there)
        for (size_t i : members_)
            if (jobs_[i].error.empty() && !streamed_[i])
                bytes[(size_t)fileDev_[i]] += align256((size_t)(window_[i].second - window_[i].first) * (size_t)jobs_[i].f.head.channels * sizeof(float));
        std::vector<char *> bases((size_t)ndev_, nullptr);
        for (int d = 0; d < ndev_; d++)
            if (bytes[(size_t)d]) bases[(size_t)d] = (char *)arena_(d, bytes[(size_t)d]);
        for (size_t gi = 0; gi < groups_.size(); gi++) {
            Group &g = groups_[gi];
            char *&base = bases[(size_t)g.dev];
            const size_t x = g.ns * g.maxF * g.ch * g.N * sizeof(float), q = g.ns * g.maxF;
            g.in = (uint8_t *)base; base += align256(q * g.frameBytes);
            g.off = nullptr;
            if (g.symbols) { g.off = (uint32_t *)base; base += align256(g.ns * (g.maxF + 1) * 4); }
            g.out = nullptr;
            if (!g.mapped) { g.out = (float *)base; base += align256(x); }
            g.pg = (float *)base; base += align256(q * 4);
            g.pp = (int *)base; base += align256(q * 4);
            g.pt = (int *)base; base += align256(q * 4);
            g.tr = (uint8_t *)base; base += align256(q);
            std::memset(g.pg, 0, q * 4);                        // padded frames: no post-filter, not transient
            std::memset(g.pp, 0, q * 4);
            std::memset(g.pt, 0, q * 4);
            std::memset(g.tr, 0, q);
            // ~24 MB of input per piece, but never fewer streams than decoding threads: long streams finish in rounds
            // of `threads` files, and the GPU's post-filter (one sequential wave per channel) wants them together.
            // (Smaller pieces of long streams -- 4 or 8 streams, more slice chains side by side -- were measured in
            // round 4: -5 % ... +9 % of the job's wall time, profiles/r04_v_piece_ab.txt; what the job waited for was
            // the serial upload -> kernels -> download of every slice, now pipelined inside the library call.)
            size_t per = std::max<size_t>((size_t)std::max(1, threads_), pieceBytes_ / std::max<size_t>(1, g.maxF * g.frameBytes));
            if (longPieceStreams_ > 0 && g.maxF * g.frameBytes * longPieceStreams_ > (kSliceBytes * 3) / 2) per = longPieceStreams_;   // (measurement switch)
            for (size_t k0 = 0; k0 < g.ns; k0 += per) {
                pieces_.emplace_back();
                Piece &p = pieces_.back();
                p.group = gi;
                p.k0 = k0;
                p.k1 = std::min(g.ns, k0 + per);
                // time slices: multiples of 64 frames (the synthesis kernels' in-wave carry chains then restart at the
                // same frames as in one call over the whole length: bit-identical results whatever the slicing)
                const size_t frame_bytes = (p.k1 - p.k0) * g.frameBytes;
                p.sliceLen = g.maxF;
                if (g.maxF * frame_bytes > (kSliceBytes * 3) / 2) p.sliceLen = std::max<size_t>(64, (kSliceBytes / frame_bytes) & ~(size_t)63);
                p.nslices = (g.maxF + p.sliceLen - 1) / p.sliceLen;
            }
        }
        for (size_t i : members_)
            if (jobs_[i].error.empty() && !streamed_[i]) {
                char *&base = bases[(size_t)fileDev_[i]];
                hostOut_[i] = (float *)base;
                base += align256((size_t)(window_[i].second - window_[i].first) * (size_t)jobs_[i].f.head.channels * sizeof(float));
            }
        progress_ = std::vector<std::atomic<long>>(n);
        for (auto &pr : progress_) pr.store(0, std::memory_order_relaxed);
        size_t pi = 0;                                          // slots -> pieces, destination pointers
        for (size_t gi = 0; gi < groups_.size(); gi++) {
            Group &g = groups_[gi];
            for (size_t k = 0; k < g.ns; k++) {
                while (!(pieces_[pi].group == gi && k >= pieces_[pi].k0 && k < pieces_[pi].k1)) pi++;
                StreamFrames &s = sf(g.ids[k]);
                s.piece = pi;
                s.in0 = g.in + k * g.maxF * g.frameBytes;
                s.frameBytes = g.frameBytes;
                s.symbols = g.symbols;
                s.bytesMode = g.bytes;
                s.off0 = g.symbols ? g.off + k * (g.maxF + 1) : nullptr;
                s.packed = packedRecords_;
                if (s.off0) s.off0[0] = 0;
                s.tr0 = g.tr + k * g.maxF;
                s.pp0 = g.pp + k * g.maxF;
                s.pt0 = g.pt + k * g.maxF;
                s.pg0 = g.pg + k * g.maxF;
                s.progress = &progress_[g.ids[k]];
                s.sliceLen = pieces_[pi].nslices > 1 ? (long)pieces_[pi].sliceLen : 0;
                if (s.plan.size() > 1) pieces_[pi].anyMore = true;
            }
        }
        for (Piece &p : pieces_)
            if (p.nslices > 1 || p.anyMore || groups_[p.group].bytes) p.state.assign(nyq_celt_state_floats(p.k1 - p.k0, groups_[p.group].ch), 0.f);
    }

    // Files whose output is the decoded stream verbatim (one mono/stereo stream, identity mapping, unit gain) are
    // copied out slice by slice as the GPU delivers them (a later segment is appended when it exists);
    // window_[i] = [first, last) sample of the stream that belongs to the file after pre-skip / end trimming.
    void planOutput() {
        for (size_t i : members_) {
            if (!jobs_[i].error.empty()) continue;
            const FileJob &job = jobs_[i];
            const OpusHead &head = job.f.head;
            const int ch = head.channels;
            const int64_t decoded = job.subs[0].samples;
            const int64_t endSample = job.f.lastGranule >= 0 ? std::min<int64_t>(decoded, job.f.lastGranule) : decoded;
            const int64_t total = std::max<int64_t>(0, endSample - head.preSkip);
            window_[i] = {head.preSkip, head.preSkip + total};
            streamed_[i] = ch <= 2 && job.subs.size() == 1 && job.subs[0].channels == ch && head.outputGainQ8 == 0 &&
                           head.mapping[0] == 0 && (ch == 1 || head.mapping[1] == 1);
            if (!streamed_[i]) fileDev_[i] = (int)(nmapped_++ % (size_t)ndev_);
        }
    }

    // Device memory for the interleaved output of every mapped file (opus_multistream_decoder.c:305-331 on the device): one
    // region per device, the files of that device side by side; zeroed when a file has silent channels (mapping 255)
    void placeOutput() {
        std::vector<size_t> bytes((size_t)ndev_, 0);
        for (size_t i : members_) {
            if (!jobs_[i].error.empty() || streamed_[i]) continue;
            const size_t b = align256((size_t)(window_[i].second - window_[i].first) * (size_t)jobs_[i].f.head.channels * sizeof(float)) + 256;
            bytes[(size_t)fileDev_[i]] += b;
        }
        std::vector<char *> base((size_t)ndev_, nullptr);
        for (int d = 0; d < ndev_; d++)
            if (bytes[(size_t)d]) base[(size_t)d] = (char *)devArena_(d, bytes[(size_t)d]);
        for (size_t i : members_) {
            if (!jobs_[i].error.empty() || streamed_[i]) continue;
            const OpusHead &head = jobs_[i].f.head;
            const size_t b = align256((size_t)(window_[i].second - window_[i].first) * (size_t)head.channels * sizeof(float)) + 256;
            devOut_[i] = (float *)base[(size_t)fileDev_[i]];
            base[(size_t)fileDev_[i]] += b;
            bool silent = false;
            for (int c = 0; c < head.channels; c++) silent = silent || head.mapping[c] == 255;
            if (silent && nyq_device_zero((nyq_ctx *)ctxs_[(size_t)fileDev_[i] * (size_t)feedersPerDev_], devOut_[i], b) != NYQ_OK)
                throw std::runtime_error("libnyq_imdct: cannot clear a file's output buffer");
        }
    }

    // The output record of elementary stream `flat` for a call whose first sample is stream sample t0: which of the file's
    // channel slots its one or two channels go to (the FIRST output channel that names them; further ones are copied on the
    // device when the file is finished), the file's sample window and the header gain.
    nyq_out_desc recordOf(size_t flat, int64_t t0) const {
        const size_t i = fileOf_[flat];
        const OpusHead &head = jobs_[i].f.head;
        const int sub = (int)(flat - firstSub_[i]);
        nyq_out_desc d;
        d.base = devOut_[i];
        d.first = window_[i].first;
        d.last = window_[i].second;
        d.t0 = t0;
        d.cstride = head.channels;
        d.coff[0] = d.coff[1] = -1;
        d.gain = head.outputGainQ8 == 0 ? 1.f   // OPUS_SET_GAIN, opus_decoder_clean.c:700-712
                                        : (float)std::exp(0.6931471805599453094 * (6.48814081e-4 * head.outputGainQ8));
        for (int c = head.channels - 1; c >= 0; c--) {           // (descending: the lowest output channel wins)
            const int idx = head.mapping[c];
            if (idx == 255) continue;
            int s2, sc;
            if (idx < 2 * head.coupledCount) { s2 = idx / 2; sc = idx & 1; }
            else { s2 = idx - head.coupledCount; sc = 0; }
            if (s2 == sub) d.coff[sc] = c;
        }
        return d;
    }

    // pass 1 for one file (a decoding thread)
    void decodeFile(size_t i) {
        FileJob &job = jobs_[i];
        if (!job.error.empty()) return;
        try {
            entropyDecode(job, [&](StreamFrames &s) {
                std::lock_guard<std::mutex> lk(mu_);
                offer(s.piece);
            });
        } catch (const std::exception &e) {
            job.error = e.what();
        }
        for (StreamFrames &s : job.subs) {
            const Group &g = groups_[s.group];
            const size_t have = job.error.empty() ? (size_t)s.plan[0].nframes : 0;   // a failed file plays as silence
            if (s.symbols) {                                    // padding: records of a zero head (silent frames), 32 bytes each
                for (size_t f = have; f < g.maxF; f++) {
                    std::memset(s.in0 + (size_t)s.off0[f] * 16, 0, 32);
                    s.off0[f + 1] = s.off0[f] + (uint32_t)(s.packed ? 2 : s.frameBytes / 16);
                }
            } else {
                std::memset(s.in0 + have * g.frameBytes, 0, (g.maxF - have) * g.frameBytes);
                if (s.bytesMode) std::memset(s.pp0 + have, 0, (g.maxF - have) * 4);      // (empty frames: silence)
            }
            if (!job.error.empty()) {
                std::memset(s.tr0, 0, g.maxF);
                std::memset(s.pp0, 0, g.maxF * 4);
                std::memset(s.pt0, 0, g.maxF * 4);
                std::memset(s.pg0, 0, g.maxF * 4);
            }
            frames += s.plan[0].nframes;
            s.progress->store((long)g.maxF, std::memory_order_release);   // decoded and padded to the group's length
            std::lock_guard<std::mutex> lk(mu_);
            offer(s.piece);
        }
    }

    // queue the next slice of piece `pi` if every stream of the piece has been decoded that far (call with mu_ held)
    void offer(size_t pi) {
        Piece &p = pieces_[pi];
        if (p.inFlight || p.nextSlice >= p.nslices) return;
        const Group &g = groups_[p.group];
        const long target = (long)std::min((p.nextSlice + 1) * p.sliceLen, g.maxF);
        for (size_t k = p.k0; k < p.k1; k++)
            if (progress_[g.ids[k]].load(std::memory_order_acquire) < target) return;
        p.inFlight = true;
        ready_[(size_t)g.dev].push_back(pi);
        cv_.notify_all();                                       // (the feeders of every device wait on this one condition)
    }

    // pass 2: one feeder thread = one GPU context; slices of one piece go in order (the decoder state of the piece
    // travels with them on the host), different pieces side by side
    void feederLoop(int which) {
        nyq_ctx *ctx = (nyq_ctx *)ctxs_[which];
        std::deque<size_t> &mine = ready_[(size_t)(which / feedersPerDev_)];   // the pieces of this feeder's device
        for (;;) {
            size_t pi;
            {
                std::unique_lock<std::mutex> lk(mu_);
                cv_.wait(lk, [&] { return !mine.empty() || finishedPieces_ == pieces_.size() || abort_; });
                if (mine.empty() || abort_) return;
                pi = mine.front();
                mine.pop_front();
            }
            Piece &p = pieces_[pi];
            const Group &g = groups_[p.group];
            const size_t myslice = p.nextSlice;
            // every slice that is ready goes in this one call: when the GPU side is what the job waits for, the slices pile
            // up, and inside one call the library pipelines them (upload / kernels / download of consecutive time windows)
            size_t m = 1;
            {
                long avail = (long)g.maxF;
                for (size_t k = p.k0; k < p.k1; k++) avail = std::min(avail, progress_[g.ids[k]].load(std::memory_order_acquire));
                while (myslice + m < p.nslices && (long)std::min((myslice + m + 1) * p.sliceLen, g.maxF) <= avail) m++;
            }
            const size_t f0 = myslice * p.sliceLen, len = std::min(m * p.sliceLen, g.maxF - f0);
            const size_t so = p.k0 * g.maxF + f0;         // first stream of the piece, first frame of the slices
            const bool last = myslice + m == p.nslices;
            bool released = false, ok = false;
            const auto c0 = std::chrono::steady_clock::now();
            try {
                int rc;
                // every stream of a mapped piece writes its samples of this slice straight into its file's interleaved
                // device buffer: channel slot, pre-skip / end trim and header gain are in the record
                std::vector<nyq_out_desc> desc(g.mapped ? p.k1 - p.k0 : 0);
                for (size_t k = p.k0; k < p.k1 && g.mapped; k++) desc[k - p.k0] = recordOf(g.ids[k], (int64_t)(f0 * g.N));
                const uint8_t *in = g.symbols ? g.in + p.k0 * g.maxF * g.frameBytes : g.in + so * g.frameBytes;   // (packed: the streams' bases)
                float *out = g.mapped ? nullptr : g.out + so * g.ch * g.N, *state = p.state.empty() ? nullptr : p.state.data();
                if (g.bytes) {
                    g_deviceEntropyFrames.fetch_add((long)((p.k1 - p.k0) * len), std::memory_order_relaxed);
                    rc = nyq_celt_bytes_to_pcm_mapped(ctx, g.LM, g.in + so * g.frameBytes, reinterpret_cast<const unsigned *>(g.pp + so), out,
                                                      g.mapped ? desc.data() : nullptr, state, p.k1 - p.k0, len, g.ch, g.maxF);
                    // what the host stage reports by throwing: a frame that read past its end (the entropy stage's state, which
                    // travels at the end of the piece's, counts them per stream)
                    const size_t nsP = p.k1 - p.k0, es = sizeof(nyq_ent_state) / sizeof(float);
                    for (size_t k = 0; rc == NYQ_OK && state && k < nsP; k++) {
                        nyq_ent_state one;
                        std::memcpy(&one, state + p.state.size() - (nsP - k) * es, sizeof one);
                        if (one.errors == 0) continue;
                        std::lock_guard<std::mutex> lk(mu_);
                        std::string &err = jobs_[fileOf_[g.ids[p.k0 + k]]].error;
                        if (err.empty()) err = "CELT frame failed to decode";
                    }
                } else if (g.symbols) {
                    rc = packedRecords_ ? nyq_celt_symbols_packed_to_pcm_mapped(ctx, g.LM, in, g.off + p.k0 * (g.maxF + 1) + f0, g.maxF * g.frameBytes,
                                                                                g.tr + so, g.pp + so, g.pg + so, g.pt + so, out,
                                                                                g.mapped ? desc.data() : nullptr, state, p.k1 - p.k0, len, g.ch, g.maxF)
                                        : nyq_celt_symbols_to_pcm_mapped(ctx, g.LM, g.in + so * g.frameBytes, g.tr + so, g.pp + so, g.pg + so, g.pt + so, out,
                                                                         g.mapped ? desc.data() : nullptr, state, p.k1 - p.k0, len, g.ch, g.maxF);
                } else if (g.mapped) {
                    rc = nyq_celt_frames_to_pcm_mapped(ctx, g.LM, reinterpret_cast<const float *>(in), g.tr + so, g.pp + so, g.pg + so, g.pt + so,
                                                       nullptr, desc.data(), state, p.k1 - p.k0, len, g.ch, g.maxF);
                } else {
                    rc = nyq_celt_frames_to_pcm_window(ctx, g.LM, reinterpret_cast<const float *>(in), g.tr + so, g.pp + so, g.pg + so, g.pt + so,
                                                       out, state, p.k1 - p.k0, len, g.ch, g.maxF);
                }
                if (rc != NYQ_OK) throw std::runtime_error(std::string("libnyq_imdct: ") + nyq_last_error(ctx));
                // the GPU part of the slice is done: the piece's next slice may start (its kernels only need the
                // state, which is back on the host) while this thread copies samples out
                if (!last) {
                    std::lock_guard<std::mutex> lk(mu_);
                    p.nextSlice += m;
                    p.inFlight = false;
                    released = true;
                    offer(pi);
                }
                {
                    std::unique_lock<std::mutex> lk(mu_);
                    cvAppend_.wait(lk, [&] { return p.appendTurn == myslice; });
                }
                handOver(p, f0, len);
                {
                    std::lock_guard<std::mutex> lk(mu_);
                    p.appendTurn = myslice + m;
                }
                cvAppend_.notify_all();
                if (last && p.anyMore) keepStates(ctx, p);
                ok = true;
            } catch (const std::exception &e) {
                std::lock_guard<std::mutex> lk(mu_);
                if (gpuError_.empty()) gpuError_ = e.what();
            } catch (...) {
                std::lock_guard<std::mutex> lk(mu_);
                if (gpuError_.empty()) gpuError_ = "unknown exception in a feeder thread";
            }
            gpuBusy_[(size_t)which] += std::chrono::duration<double>(std::chrono::steady_clock::now() - c0).count();
            bool finishFiles = false;
            {
                // Every earlier slice of the piece has handed its samples over before this one is closed -- also on the
                // error path, where the hand-over above was skipped: nothing may touch the files' sample vectors (pass 3
                // below) while an earlier slice's feeder is still appending to them.
                std::unique_lock<std::mutex> lk(mu_);
                cvAppend_.wait(lk, [&] { return p.appendTurn >= myslice; });   // (earlier slices always get there)
                if (p.appendTurn == myslice) {              // an error above skipped the hand-over: do not block later slices
                    p.appendTurn = myslice + m;
                    cvAppend_.notify_all();
                }
                finishFiles = last && ok && gpuError_.empty();   // (after a GPU failure run() throws: no file is finished)
            }
            if (finishFiles) finishFilesOf(p, which);
            std::lock_guard<std::mutex> lk(mu_);
            if (!released) {
                p.nextSlice += m;
                p.inFlight = false;
                if (p.nextSlice == p.nslices) {
                    if (++finishedPieces_ == pieces_.size()) cv_.notify_all();
                } else {
                    offer(pi);
                }
            }
        }
    }

    // hand the slice's samples to the files that take them verbatim (slices of a piece arrive here in order)
    void handOver(const Piece &p, size_t f0, size_t len) {
        const Group &g = groups_[p.group];
        if (g.mapped) return;                                   // (its samples are in the files' device buffers)
        // Once the entropy stage is over its threads' CPUs are idle and what is left of the job is this copying: a large
        // hand-over is then split over a few short-lived threads (each stream has its own destination vector).  While the
        // entropy stage runs, the copy stays on the feeder: every CPU has work.
        const size_t bytes = (p.k1 - p.k0) * len * g.N * (size_t)g.ch * sizeof(float);
        const size_t helpers = decodeDone_.load(std::memory_order_acquire) && bytes >= ((size_t)4 << 20)
                                   ? std::min<size_t>(std::min<size_t>(8, (size_t)std::max(1, threads_)), p.k1 - p.k0)
                                   : 1;
        if (helpers > 1) {
            std::atomic<bool> failed{false};
            size_t started = 0;
            {
                JoinedThreads pool;                                 // (joined at the end of this block, whatever happens)
                try {
                    for (; started < helpers; started++)
                        pool.start([this, &p, &failed, f0, len, h = started, helpers] {
                            try {
                                for (size_t k = p.k0 + h; k < p.k1; k += helpers) handOverStream(p, k, f0, len);
                            } catch (...) {
                                failed.store(true);
                            }
                        });
                } catch (const std::system_error &) {               // no more threads to be had: the rest is done here
                }
                for (size_t h = started; h < helpers; h++)
                    for (size_t k = p.k0 + h; k < p.k1; k += helpers) handOverStream(p, k, f0, len);
            }
            if (failed.load()) throw std::runtime_error("the hand-over of a slice's samples failed (out of memory)");
            return;
        }
        for (size_t k = p.k0; k < p.k1; k++) handOverStream(p, k, f0, len);
    }
    void handOverStream(const Piece &p, size_t k, size_t f0, size_t len) {
        const Group &g = groups_[p.group];
        {
            const size_t i = fileOf_[g.ids[k]];
            if (!streamed_[i]) return;
            const int64_t n0 = (int64_t)sf(g.ids[k]).plan[0].nframes * (int64_t)g.N;   // (frames past it are padding)
            const int64_t lo = std::max<int64_t>(window_[i].first, (int64_t)(f0 * g.N));
            const int64_t hi = std::min<int64_t>(std::min<int64_t>(window_[i].second, n0), (int64_t)((f0 + len) * g.N));
            if (hi <= lo) return;
            std::vector<float> &pcm = out_[i].pcm;
            if (pcm.empty()) pcm.reserve((size_t)(window_[i].second - window_[i].first) * g.ch);
            const float *src = g.out + k * g.maxF * g.N * g.ch;
            pcm.insert(pcm.end(), src + lo * g.ch, src + hi * g.ch);
        }
    }

    // after a piece's last slice: the decoder state of every stream that continues with another segment.  The piece's
    // state is exact only for streams whose first segment fills the whole padded length; a shorter one is given
    // its own call.
    void keepStates(nyq_ctx *ctx, const Piece &p) {
        const Group &g = groups_[p.group];
        for (size_t k = p.k0; k < p.k1; k++) {
            const size_t flat = g.ids[k];
            StreamFrames &s = sf(flat);
            if (s.plan.size() < 2) continue;
            std::vector<float> st1(nyq_celt_state_floats(1, g.ch), 0.f);
            if ((size_t)s.plan[0].nframes != g.maxF) {
                std::vector<float> out1((size_t)s.plan[0].nframes * g.N * g.ch);
                const size_t nf0 = (size_t)s.plan[0].nframes;
                if ((g.bytes   ? nyq_celt_bytes_to_pcm_mapped(ctx, g.LM, s.in0, reinterpret_cast<const unsigned *>(s.pp0), out1.data(), nullptr, st1.data(), 1,
                                                              nf0, g.ch, g.maxF)
                     : g.symbols ? nyq_celt_symbols_packed_to_pcm_mapped(ctx, g.LM, s.in0, s.off0, 0, s.tr0, s.pp0, s.pg0, s.pt0, out1.data(), nullptr,
                                                                       st1.data(), 1, nf0, g.ch, g.maxF)
                               : nyq_celt_frames_to_pcm(ctx, g.LM, reinterpret_cast<const float *>(s.in0), s.tr0, s.pp0, s.pg0, s.pt0, out1.data(),
                                                        st1.data(), 1, nf0, g.ch)) != NYQ_OK)
                    throw std::runtime_error(std::string("libnyq_imdct: ") + nyq_last_error(ctx));
            } else {
                stateOfStream(p.state.data(), p.k1 - p.k0, g.ch, k - p.k0, st1.data());
            }
            stateOf_[flat] = std::move(st1);
        }
    }

    // pass 3 right away for every file all of whose streams are through (and that has no later segment): what is left
    // of it overlaps the entropy stage of the files still being decoded
    void finishFilesOf(const Piece &p, int which) {
        const Group &g = groups_[p.group];
        for (size_t k = p.k0; k < p.k1; k++) {
            const size_t i = fileOf_[g.ids[k]];
            if (subsLeft_[i].fetch_sub(1, std::memory_order_acq_rel) != 1 || !jobs_[i].error.empty()) continue;
            bool simple = true;
            for (const auto &sub : jobs_[i].subs) simple = simple && sub.plan.size() == 1;
            if (!simple) continue;
            try {
                finishFile(i, which);
            } catch (const std::exception &e) {
                std::lock_guard<std::mutex> lk(mu_);
                if (gpuError_.empty()) gpuError_ = e.what();
            }
        }
    }

    // Later segments (a stream that changes its frame size, typically a short closing frame): round r takes
    // segment r of every stream that has one, batched over the streams of equal shape (channels, frame size,
    // frame count -- no padding, so the decoder state that comes back is exact) with their states gathered
    // into the batch layout and scattered back.
    void laterSegments() {
        const size_t n = sfp_.size();
        for (size_t r = 1;; r++) {
            // (device, channels, frame size, frame count): a stream's later segments run on the device that holds its
            // group -- no stream's data ever visits another device
            std::map<std::tuple<int, int, int, long, int>, std::vector<size_t>> shapes;
            for (size_t i = 0; i < n; i++)
                if (sf(i).later.size() >= r && !stateOf_[i].empty())   // (no state: the file failed in pass 1)
                    shapes[std::make_tuple(groups_[sf(i).group].dev, sf(i).channels, sf(i).later[r - 1].LM, sf(i).later[r - 1].nframes,
                                           (groups_[sf(i).group].mapped ? 1 : 0) | (sf(i).bytesMode ? 2 : 0))]
                        .push_back(i);
            if (shapes.empty()) break;
            for (const auto &kv : shapes) {
                nyq_ctx *ctx = (nyq_ctx *)ctxs_[(size_t)std::get<0>(kv.first) * (size_t)feedersPerDev_];
                const int ch = std::get<1>(kv.first), LM = std::get<2>(kv.first);
                const size_t nf = (size_t)std::get<3>(kv.first), N = (size_t)120 << LM;
                const bool mapped = (std::get<4>(kv.first) & 1) != 0, byBytes = (std::get<4>(kv.first) & 2) != 0;
                const std::vector<size_t> &ids = kv.second;
                const size_t ns = ids.size(), per = nf * ch * N, bslot = nyq_celt_byte_slot();
                std::vector<float> freq(byBytes ? 0 : ns * per), pcm(mapped ? 0 : ns * per), pg(ns * nf), state(nyq_celt_state_floats(ns, ch));
                std::vector<uint8_t> fbytes(byBytes ? ns * nf * bslot : 0);
                std::vector<uint32_t> fwords(byBytes ? ns * nf : 0);
                std::vector<nyq_out_desc> desc(mapped ? ns : 0);
                std::vector<int> pp(ns * nf), pt(ns * nf);
                std::vector<uint8_t> tr(ns * nf);
                for (size_t k = 0; k < ns; k++) {
                    const Segment &sg = sf(ids[k]).later[r - 1];
                    if (byBytes) {
                        std::memcpy(&fbytes[k * nf * bslot], sg.bytes.data(), nf * bslot);
                        std::memcpy(&fwords[k * nf], sg.words.data(), nf * sizeof(uint32_t));
                    } else {
                        std::memcpy(&freq[k * per], sg.freq.data(), per * sizeof(float));
                        std::memcpy(&tr[k * nf], sg.transient.data(), nf);
                        std::memcpy(&pp[k * nf], sg.pfPitch.data(), nf * sizeof(int));
                        std::memcpy(&pt[k * nf], sg.pfTapset.data(), nf * sizeof(int));
                        std::memcpy(&pg[k * nf], sg.pfGain.data(), nf * sizeof(float));
                    }
                    stateIntoBatch(state.data(), ns, ch, k, stateOf_[ids[k]].data());
                    if (mapped) {                                   // samples decoded before this segment: where it starts in the stream
                        const StreamFrames &st = sf(ids[k]);
                        int64_t t0 = (int64_t)st.plan[0].nframes * (int64_t)((size_t)120 << st.plan[0].LM);
                        for (size_t q = 0; q + 1 < r; q++) t0 += (int64_t)st.later[q].nframes * (int64_t)((size_t)120 << st.later[q].LM);
                        desc[k] = recordOf(ids[k], t0);
                    }
                }
                const int rc = byBytes ? nyq_celt_bytes_to_pcm_mapped(ctx, LM, fbytes.data(), fwords.data(), mapped ? nullptr : pcm.data(),
                                                                      mapped ? desc.data() : nullptr, state.data(), ns, nf, ch, nf)
                               : mapped ? nyq_celt_frames_to_pcm_mapped(ctx, LM, freq.data(), tr.data(), pp.data(), pg.data(), pt.data(), nullptr,
                                                                     desc.data(), state.data(), ns, nf, ch, nf)
                                      : nyq_celt_frames_to_pcm(ctx, LM, freq.data(), tr.data(), pp.data(), pg.data(), pt.data(), pcm.data(),
                                                               state.data(), ns, nf, ch);
                if (rc != NYQ_OK) throw std::runtime_error(std::string("libnyq_imdct: ") + nyq_last_error(ctx));
                for (size_t k = 0; k < ns && byBytes; k++) {                 // frames in error fail their file (as the host stage's exception does)
                    nyq_ent_state one;
                    std::memcpy(&one, state.data() + state.size() - (ns - k) * (sizeof one / sizeof(float)), sizeof one);
                    if (one.errors && jobs_[fileOf_[ids[k]]].error.empty()) jobs_[fileOf_[ids[k]]].error = "CELT frame failed to decode";
                }
                for (size_t k = 0; k < ns; k++) {
                    const size_t i = ids[k];
                    if (!mapped) laterPcm_[i].insert(laterPcm_[i].end(), &pcm[k * per], &pcm[k * per] + per);
                    stateOfStream(state.data(), ns, ch, k, stateOf_[i].data());
                    frames += (long)nf;
                }
            }
        }
    }

    // pass 3 for one file.  A file that is one identity-mapped stream at unit gain has received its first segment slice by
    // slice (handOver); what its later segments add is appended here, a contiguous range.  Every other file -- several
    // elementary streams, a permuted or partly silent mapping, a header gain -- was written by the kernels themselves into
    // its interleaved layout in device memory (channel mapping opus_multistream_decoder.c:305-331, pre-skip / end trim,
    // OPUS_SET_GAIN opus_decoder_clean.c:700-712: recordOf): output channels that repeat a decoded channel are copied on
    // the device, then ONE download.  No per-sample work on the host.
    void finishFile(size_t i, int which) {
        finished_[i] = 1;
        const FileJob &job = jobs_[i];
        const OpusHead &head = job.f.head;
        DecodedStream &d = out_[i];
        const int ch = head.channels;
        d.channels = ch;
        d.preSkip = head.preSkip;
        for (const auto &sub : job.subs) {
            d.frames += sub.nframes;
            d.transientFrames += sub.transientCount;
        }
        const int64_t a = window_[i].first, b = window_[i].second, total = b - a;
        d.totalSamples = total;
        if (streamed_[i]) {                                 // the first segment went out slice by slice
            const StreamFrames &s = sf(firstSub_[i]);
            const int64_t n0 = (int64_t)s.plan[0].nframes * (int64_t)((size_t)120 << s.plan[0].LM);
            if (b > n0) {                                   // what the later segments add
                const float *src1 = laterPcm_[firstSub_[i]].data();
                d.pcm.insert(d.pcm.end(), src1 + (std::max(a, n0) - n0) * ch, src1 + (b - n0) * ch);
            }
            return;
        }
        if (total == 0) return;
        nyq_ctx *ctx = (nyq_ctx *)ctxs_[(size_t)which];
        std::lock_guard<std::mutex> lk(ctxMu_[(size_t)which]);
        // an output channel that names a decoded channel another (lower) output channel already carries
        std::vector<int> firstOf(256, -1);
        for (int c = 0; c < ch; c++) {
            const int idx = head.mapping[c];
            if (idx == 255) continue;
            if (firstOf[(size_t)idx] < 0) firstOf[(size_t)idx] = c;
            else if (nyq_device_dup_channel(ctx, devOut_[i], ch, firstOf[(size_t)idx], c, (size_t)total) != NYQ_OK)
                throw std::runtime_error(std::string("libnyq_imdct: ") + nyq_last_error(ctx));
        }
        if (nyq_device_download(ctx, hostOut_[i], devOut_[i], (size_t)total * (size_t)ch * sizeof(float)) != NYQ_OK)
            throw std::runtime_error(std::string("libnyq_imdct: ") + nyq_last_error(ctx));
        d.pcm.assign(hostOut_[i], hostOut_[i] + (size_t)total * ch);
    }

    std::vector<FileJob> &jobs_;
    std::vector<DecodedStream> &out_;
    const std::vector<size_t> &members_;
    void *const *ctxs_;
    const int ndev_, feedersPerDev_, nfeeders_, threads_;
    const std::function<void *(int, size_t)> &arena_;
    const std::function<void *(int, size_t)> &devArena_;

    std::vector<StreamFrames *> sfp_;                  // flattened elementary streams
    std::vector<size_t> firstSub_, fileOf_;            // file -> first flat index; flat index -> file
    std::vector<Group> groups_;
    std::vector<Piece> pieces_;
    std::vector<std::atomic<long>> progress_;          // per flat stream: frames of the first segment decoded (and padded)
    std::vector<std::vector<float>> stateOf_;          // decoder state of streams that continue (nstreams = 1 layout)
    std::vector<std::vector<float>> laterPcm_;         // segments 1.. of the streams that have them
    std::vector<char> finished_, streamed_;            // per file: pass 3 done; samples handed over slice by slice
    std::vector<std::pair<int64_t, int64_t>> window_;  // per file: [first, last) sample after trimming
    std::vector<std::atomic<int>> subsLeft_;           // per file: elementary streams still on their way through the GPU
    std::vector<int> fileDev_;                         // per mapped file: the device (index into the decoder's list) of all its streams
    std::vector<float *> devOut_;                      // per mapped file: its interleaved output in that device's memory
    std::vector<float *> hostOut_;                     // ... and where it lands in the page-locked arena on its way to the caller
    size_t nmapped_ = 0;
    std::vector<std::mutex> ctxMu_{(size_t)nfeeders_}; // per context: finishFile's device calls (a feeder owns its context otherwise)

    std::mutex mu_;                                    // scheduler: ready queue, Piece::{nextSlice, inFlight, appendTurn}
    std::condition_variable cv_, cvAppend_;
    std::vector<std::deque<size_t>> ready_;   // per device
    size_t finishedPieces_ = 0;
    bool abort_ = false;                               // run() is unwinding: feeders stop taking pieces
    std::atomic<bool> decodeDone_{false};              // the entropy stage is over: its CPUs are free for the hand-over copies
    std::string gpuError_;
    std::vector<double> gpuBusy_;
};

}  // namespace

BatchOpusDecoder::BatchOpusDecoder(int device) : BatchOpusDecoder(std::vector<int>{device}) {}

BatchOpusDecoder::BatchOpusDecoder(const std::vector<int> &devices) : devices_(devices) {
    if (devices_.empty()) throw std::runtime_error("BatchOpusDecoder: empty device list");
    if (const char *e = std::getenv("NYQ_BATCH_BYTES")) {      // read once, here: no decoding path looks at the environment
        const long long v = std::atoll(e);
        if (v > 0) stagingBudget_ = (size_t)v;
    }
    if (const char *e = std::getenv("NYQ_HOST_SYMBOLS")) symbolRecords_ = std::atoi(e) != 0;   // (A/B switch, read once like the budget)
    if (const char *e = std::getenv("NYQ_DEVICE_ENTROPY")) deviceEntropy_ = std::atoi(e) != 0; // (opt-in this round: DESIGN 4.11)
    if (const char *e = std::getenv("NYQ_BATCH_TRACE")) trace_ = std::atoi(e) != 0;
    long hostWindow = -1;                                   // (measurement switch: NYQ_OPT_HOST_WINDOW of every context)
    if (const char *e = std::getenv("NYQ_HOST_WINDOW")) hostWindow = std::atol(e);
    if (const char *e = std::getenv("NYQ_HOST_PACKED")) packedRecords_ = std::atoi(e) != 0;
    if (const char *e = std::getenv("NYQ_LONG_PIECE_STREAMS")) longPieceStreams_ = (size_t)std::max(0, std::atoi(e));
    if (const char *e = std::getenv("NYQ_PIECE_BYTES")) pieceBytes_ = (size_t)std::max<long long>(1 << 20, std::atoll(e));
    const int ndev = nyq_device_count();
    for (int d : devices_)
        if (d < 0 || d >= ndev)
            throw std::runtime_error("BatchOpusDecoder: device " + std::to_string(d) + " does not exist (" + std::to_string(ndev) + " visible)");
    arenas_.resize(devices_.size());
    devArenas_.resize(devices_.size());
    for (size_t d = 0; d < devices_.size(); d++)
        for (int k = 0; k < kFeeders; k++) {
            nyq_ctx *c = nullptr;
            if (nyq_ctx_create(&c, devices_[d]) != NYQ_OK) {
                for (void *p : ctx_) nyq_ctx_destroy((nyq_ctx *)p);
                ctx_.clear();
                throw std::runtime_error(std::string("libnyq_imdct: ") + nyq_last_error(nullptr));
            }
            if (hostWindow >= 0) (void)nyq_ctx_set_option(c, NYQ_OPT_HOST_WINDOW, hostWindow);
            ctx_.push_back(c);
            if (deviceEntropy_) {                               // the entropy stage's tables, from this library's own mode
                std::vector<uint8_t> tables(entropyTablesBytes());
                fillEntropyTables(tables.data());
                if (nyq_ctx_set_entropy_tables(c, tables.data(), tables.size()) != NYQ_OK) {
                    const std::string why = nyq_last_error(c);
                    for (void *p : ctx_) nyq_ctx_destroy((nyq_ctx *)p);
                    ctx_.clear();
                    throw std::runtime_error("libnyq_imdct: " + why);
                }
            }
        }
}

BatchOpusDecoder::~BatchOpusDecoder() {
    for (Arena &a : arenas_)
        if (a.p) (a.pinned ? nyq_host_free(a.p) : std::free(a.p));
    for (size_t d = 0; d < devArenas_.size(); d++)
        if (devArenas_[d].p) nyq_device_free((nyq_ctx *)ctx_[d * kFeeders], devArenas_[d].p);
    for (size_t k = ctx_.size(); k-- > 0;) nyq_ctx_destroy((nyq_ctx *)ctx_[k]);
}

void BatchOpusDecoder::trim(size_t keepBytes) {
    for (Arena &a : arenas_)
        if (a.p && a.bytes > keepBytes) {
            a.pinned ? nyq_host_free(a.p) : std::free(a.p);
            a = Arena();
        }
    for (size_t d = 0; d < devArenas_.size(); d++)
        if (devArenas_[d].p && devArenas_[d].bytes > keepBytes) {
            nyq_device_free((nyq_ctx *)ctx_[d * kFeeders], devArenas_[d].p);
            devArenas_[d] = Arena();
        }
    // pooled sample buffers, ascending by capacity (decodeImpl hands them out from the back: the largest first); keep
    // the largest ones that fit the same bound
    std::sort(pool_.begin(), pool_.end(), [](const std::vector<float> &a, const std::vector<float> &b) { return a.capacity() < b.capacity(); });
    size_t kept = 0, firstKept = pool_.size();
    while (firstKept > 0 && kept + pool_[firstKept - 1].capacity() * sizeof(float) <= keepBytes) kept += pool_[--firstKept].capacity() * sizeof(float);
    pool_.erase(pool_.begin(), pool_.begin() + (long)firstKept);
}

// page-locked staging memory of one device, kept from call to call (grow only); pageable memory if pinning fails
void *BatchOpusDecoder::arena(int dev, size_t bytes) {
    Arena &a = arenas_[(size_t)dev];
    if (bytes <= a.bytes) return a.p;
    if (a.p) (a.pinned ? nyq_host_free(a.p) : std::free(a.p));
    a = Arena();
    const size_t want = bytes + bytes / 8;
    a.p = nyq_host_alloc(want);
    a.pinned = a.p != nullptr;
    if (!a.p) a.p = std::aligned_alloc(4096, (want + 4095) & ~(size_t)4095);
    if (!a.p) throw std::bad_alloc();
    a.bytes = want;
    return a.p;
}

// device memory of one device for the mapped files' interleaved output, kept from call to call (grow only: a hipFree is a
// device-wide synchronisation, never on the decode path)
void *BatchOpusDecoder::deviceArena(int dev, size_t bytes) {
    Arena &a = devArenas_[(size_t)dev];
    if (bytes <= a.bytes) return a.p;
    nyq_ctx *c = (nyq_ctx *)ctx_[(size_t)dev * kFeeders];
    if (a.p) nyq_device_free(c, a.p);
    a = Arena();
    const size_t want = bytes + bytes / 8;
    a.p = nyq_device_alloc(c, want);
    if (!a.p) throw std::runtime_error("libnyq_imdct: cannot allocate device memory for the files' output");
    a.bytes = want;
    return a.p;
}

void BatchOpusDecoder::decode(const std::vector<const std::vector<uint8_t> *> &files, std::vector<DecodedStream> &out,
                              BatchStats *stats, int threads) {
    decodeImpl(files, out, nullptr, stats, threads);
}

void BatchOpusDecoder::decode(const std::vector<const std::vector<uint8_t> *> &files, const StreamSink &sink, BatchStats *stats,
                              int threads) {
    std::vector<DecodedStream> out;
    decodeImpl(files, out, &sink, stats, threads);
}

void BatchOpusDecoder::decodeImpl(const std::vector<const std::vector<uint8_t> *> &files, std::vector<DecodedStream> &out,
                                  const StreamSink *sink, BatchStats *stats, int threads) {
    const size_t nfiles = files.size();
    out.assign(nfiles, DecodedStream());
    std::vector<FileJob> jobs(nfiles);
    if (threads <= 0) threads = usableHostThreads();
    const auto t0 = std::chrono::steady_clock::now();
    // pass 0: scan
    parallelFor(nfiles, threads, [&](size_t i) {
        try {
            scanFile(*files[i], jobs[i]);
        } catch (const std::exception &e) {
            jobs[i].error = e.what();
        }
    });
    // Memory bound: the page-locked group buffers hold freq[] and PCM of every stream of a batch, so a big job
    // (BASELINE config 4: 1000 streams of a 224 s file = 86 GB of coefficients) runs as consecutive sub-batches
    // of at most NYQ_BATCH_BYTES (default 12 GiB) of staging memory; files keep their order.
    const size_t budget = stagingBudget_;
    std::vector<std::vector<size_t>> batches(1);
    {
        std::map<std::pair<int, int>, std::pair<size_t, size_t>> shape;   // (channels, LM) -> (streams, longest)
        auto add = [&](const FileJob &job) {
            for (const auto &sub : job.subs) {
                auto &e = shape[{sub.channels, sub.plan[0].LM}];
                e.first++;
                e.second = std::max(e.second, (size_t)sub.plan[0].nframes);
            }
        };
        auto estimate = [&]() {
            size_t bytes = 0;
            for (const auto &kv : shape) {
                const size_t pcm = (size_t)kv.first.first * ((size_t)120 << kv.first.second) * sizeof(float);
                const size_t in = deviceEntropy_ && kv.first.first <= 2 ? nyq_celt_byte_slot()
                                  : symbolRecords_ && kv.first.first <= 2 ? nyq_celt_symbol_bytes_lm(kv.first.first, kv.first.second) : pcm;
                bytes += kv.second.first * kv.second.second * (in + pcm);
            }
            return bytes;
        };
        for (size_t i = 0; i < nfiles; i++) {
            if (!jobs[i].error.empty()) continue;
            add(jobs[i]);
            if (estimate() > budget && !batches.back().empty()) {
                // close the sub-batch on a multiple of the thread count: the decoding threads take files in rounds,
                // and a last round of one long file would leave the other threads idle
                std::vector<size_t> &cur = batches.back();
                std::vector<size_t> carry;
                const size_t keep = cur.size() >= (size_t)threads ? cur.size() - cur.size() % (size_t)threads : cur.size();
                carry.assign(cur.begin() + (long)keep, cur.end());
                cur.resize(keep);
                batches.emplace_back(carry);
                shape.clear();
                for (size_t j : carry) add(jobs[j]);
                add(jobs[i]);
            }
            batches.back().push_back(i);
        }
    }
    double cpuSecs = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count(), tailSecs = 0, busySecs = 0;
    long totalFrames = 0;
    const std::function<void *(int, size_t)> arenaFn = [this](int dev, size_t bytes) { return arena(dev, bytes); };
    const std::function<void *(int, size_t)> devArenaFn = [this](int dev, size_t bytes) { return deviceArena(dev, bytes); };
    auto settle = [&](size_t i) {
        if (!jobs[i].error.empty()) {
            out[i].error = jobs[i].error;
            out[i].pcm.clear();                            // no partial audio of a file that failed half way
        }
    };
    for (const std::vector<size_t> &members : batches) {
        if (sink)                                          // sample buffers out of the pool (capacity kept from earlier calls)
            for (size_t i : members)
                if (!pool_.empty()) {
                    out[i].pcm = std::move(pool_.back());
                    pool_.pop_back();
                    out[i].pcm.clear();
                }
        SubBatch sb(jobs, out, members, ctx_.data(), (int)devices_.size(), kFeeders, threads, arenaFn, devArenaFn);
        sb.symbolRecords_ = symbolRecords_;
        sb.deviceEntropy_ = deviceEntropy_;
        sb.packedRecords_ = packedRecords_;
        sb.pieceBytes_ = pieceBytes_;
        sb.longPieceStreams_ = longPieceStreams_;
        sb.trace_ = trace_;
        sb.run();
        cpuSecs += sb.cpuSeconds;
        tailSecs += sb.tailSeconds;
        busySecs += sb.busySeconds;
        totalFrames += sb.frames.load();
        if (sink)
            for (size_t i : members) {
                settle(i);
                (*sink)(i, out[i]);
                if (out[i].pcm.capacity()) pool_.push_back(std::move(out[i].pcm));
                out[i] = DecodedStream();
            }
    }
    std::vector<char> inBatch(nfiles, 0);
    for (const std::vector<size_t> &members : batches)
        for (size_t i : members) inBatch[i] = 1;
    for (size_t i = 0; i < nfiles; i++) {
        if (!sink) {
            settle(i);
        } else if (!inBatch[i]) {                          // a file that failed the scan is in no sub-batch
            settle(i);
            (*sink)(i, out[i]);
        }
    }
    if (stats) {
        stats->wallSeconds = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
        stats->cpuSeconds = cpuSecs;
        stats->gpuSeconds = tailSecs;
        stats->gpuBusySeconds = busySecs;
        stats->frames = totalFrames;
        stats->threads = threads;
    }
}

long deviceEntropyFrames() { return g_deviceEntropyFrames.load(); }

}  // namespace nyq_host
