// oracle/ref_capture.cpp -- runs the REAL reference decoder (NyquistIO::Load, built from the
// reference's own sources by oracle/Makefile with the capture tap of oracle/tap/) on an Opus file
// and writes what the IMDCT hot path saw.  TEST INFRASTRUCTURE ONLY; produces fixtures.
//
//   ref_capture <file.opus> <out.bin> [max_frames] [calls]
//
// With a fourth argument, <out.bin>.calls is written too: a digest of EVERY clt_mdct_backward call of the file, any frame
// size, in call order -- int32 ncalls, then per call int32 shift, stride, n2; float64 sum, sum of squares; float32 the eight
// coefficients k * n2 / 16 (the call's input is a block of the frame's freq[]: the entropy stage's whole output, digested).
//
// out.bin (little endian): int32 magic 'NYQC', channels, frames, total_calls_seen, float32
// checksum (sum of decoded samples, as examples/src/Main.cpp:137-154), int64 decoded sample count,
// then per frame: uint8 transient, and per channel freq[960], out[960+60] (for a transient frame
// `freq` is the interleaved 960-coefficient frame exactly as the decoder holds it and `out` is the
// channel's out_syn[0..1020) after the 8th block).  Only LM=3 (20 ms) stereo/mono CELT frames are
// grouped; anything else stops the capture.
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <vector>

#include "libnyquist/Decoders.h"

extern "C" {
typedef struct {
    const float *in_ptr;
    float *out_ptr;
    int shift, stride, n2;
    float *in_copy;
    float *out_copy;
} nyq_tap_call;
void nyq_tap_start(long max_calls);
long nyq_tap_count(void);
const nyq_tap_call *nyq_tap_get(long i);
typedef struct {
    float *y;
    int T0, T1, N, tapset0, tapset1;
    float g0, g1;
    float *hist;
} nyq_comb_call;
void nyq_comb_tap_start(long max_calls);
long nyq_comb_tap_count(void);
const nyq_comb_call *nyq_comb_tap_get(long i);
}

int main(int argc, char **argv) {
    if (argc < 3) { std::fprintf(stderr, "usage: ref_capture file.opus out.bin [max_frames]\n"); return 2; }
    const long max_frames = argc > 3 ? std::atol(argv[3]) : 64;
    nyq_tap_start(max_frames * 16 + 64);
    nyq_comb_tap_start(max_frames * 4 + 64);
    nqr::NyquistIO loader;
    nqr::AudioData data;
    loader.Load(&data, std::string(argv[1]));
    float sum = 0;
    for (float v : data.samples) sum += v;
    const int ch = data.channelCount;
    const long ncalls = nyq_tap_count();
    std::vector<uint8_t> flags;
    std::vector<float> payload;
    long i = 0, frames = 0;
    while (i < ncalls && frames < max_frames) {
        const nyq_tap_call *c = nyq_tap_get(i);
        if (c->shift == 0 && c->stride == 1) {
            if (i + ch > ncalls) break;
            flags.push_back(0);
            for (int k = 0; k < ch; k++) {
                const nyq_tap_call *q = nyq_tap_get(i + k);
                if (q->shift != 0) { i = ncalls; break; }
                payload.insert(payload.end(), q->in_copy, q->in_copy + 960);
                payload.insert(payload.end(), q->out_copy, q->out_copy + 1020);
            }
            i += ch;
            frames++;
        } else if (c->shift == 3 && c->stride == 8) {
            if (i + 8 * ch > ncalls) break;
            flags.push_back(1);
            // calls come block-major: b = 0..7, inside each block channel 0..ch-1
            for (int k = 0; k < ch; k++) {
                std::vector<float> X(960), out(1020);
                for (int b = 0; b < 8; b++) {
                    const nyq_tap_call *q = nyq_tap_get(i + b * ch + k);
                    for (int j = 0; j < 120; j++) X[b + 8 * j] = q->in_copy[j];
                    // block b's out buffer covers out_syn[120b .. 120b+180)
                    std::memcpy(&out[120 * b], q->out_copy, sizeof(float) * 180);
                }
                payload.insert(payload.end(), X.begin(), X.end());
                payload.insert(payload.end(), out.begin(), out.end());
            }
            i += 8 * ch;
            frames++;
        } else {
            break;   // other frame sizes: not grouped by this tool
        }
    }
    FILE *f = std::fopen(argv[2], "wb");
    if (!f) return 3;
    int32_t hdr[4] = {0x4351594e, ch, (int32_t)frames, (int32_t)ncalls};
    int64_t nsamp = (int64_t)data.samples.size();
    std::fwrite(hdr, sizeof hdr, 1, f);
    std::fwrite(&sum, sizeof sum, 1, f);
    std::fwrite(&nsamp, sizeof nsamp, 1, f);
    std::fwrite(flags.data(), 1, flags.size(), f);
    std::fwrite(payload.data(), sizeof(float), payload.size(), f);
    std::fclose(f);
    // <out>.post: post-filter calls (2 per channel per LM=3 frame: N=120 then N=840) and the decoded PCM
    //   int32 ncomb, channels; per call: int32 T0,T1,N,tapset0,tapset1; float g0,g1; int32 has_hist; [1088 floats]
    //   then int64 nsamples and the interleaved float PCM of AudioData::samples
    {
        std::string pp = std::string(argv[2]) + ".post";
        FILE *g = std::fopen(pp.c_str(), "wb");
        if (!g) return 4;
        int32_t n = (int32_t)nyq_comb_tap_count(), cc = ch;
        std::fwrite(&n, 4, 1, g);
        std::fwrite(&cc, 4, 1, g);
        for (long k = 0; k < n; k++) {
            const nyq_comb_call *q = nyq_comb_tap_get(k);
            int32_t iv[5] = {q->T0, q->T1, q->N, q->tapset0, q->tapset1};
            float fv[2] = {q->g0, q->g1};
            int32_t hh = q->hist ? 1 : 0;
            std::fwrite(iv, sizeof iv, 1, g);
            std::fwrite(fv, sizeof fv, 1, g);
            std::fwrite(&hh, 4, 1, g);
            if (hh) std::fwrite(q->hist, sizeof(float), 1088, g);
        }
        std::fwrite(&nsamp, sizeof nsamp, 1, g);
        std::fwrite(data.samples.data(), sizeof(float), data.samples.size(), g);
        std::fclose(g);
    }
    if (argc > 4) {
        std::string cp = std::string(argv[2]) + ".calls";
        FILE *g = std::fopen(cp.c_str(), "wb");
        if (!g) return 5;
        int32_t n = (int32_t)ncalls;
        std::fwrite(&n, 4, 1, g);
        for (long k = 0; k < ncalls; k++) {
            const nyq_tap_call *q = nyq_tap_get(k);
            int32_t iv[3] = {q->shift, q->stride, q->n2};
            double acc[2] = {0, 0};
            for (int j = 0; j < q->n2; j++) { acc[0] += q->in_copy[j]; acc[1] += (double)q->in_copy[j] * q->in_copy[j]; }
            float pick[8];
            for (int j = 0; j < 8; j++) pick[j] = q->in_copy[j * q->n2 / 16];
            std::fwrite(iv, sizeof iv, 1, g);
            std::fwrite(acc, sizeof acc, 1, g);
            std::fwrite(pick, sizeof pick, 1, g);
        }
        std::fclose(g);
    }
    std::printf("%s: channels %d, decoded samples %lld, sum %f, imdct calls recorded %ld, frames written %ld\n", argv[1], ch,
                (long long)nsamp, sum, ncalls, frames);
    return 0;
}
