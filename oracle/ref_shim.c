/*
 * oracle/ref_shim.c -- thin accessors over the REAL reference hot path.
 * TEST INFRASTRUCTURE ONLY.  Compiled by oracle/Makefile together with the
 * reference's own sources (read in place from /root/reference, never copied)
 * into oracle/_ref/libnyq_ref.so.  It exposes the static 48 kHz mode's tables
 * and plain-pointer wrappers so that Python (ctypes) can drive
 * clt_mdct_backward / opus_ifft without knowing the CELTMode struct layout.
 *
 * Reference entry points used (relative to /root/reference):
 *   third_party/opus/celt/modes.c:256-259      opus_custom_mode_create (static mode lookup)
 *   third_party/opus/celt/mdct.h:66-68         clt_mdct_backward
 *   third_party/opus/celt/mdct.c:258-265       clt_mdct_backward_B1_C2
 *   third_party/opus/celt/kiss_fft.h:131       opus_ifft
 *   third_party/opus/celt/kiss_fft.c:749-771   test_opus_ifft (exported by the reference itself)
 */
#include <string.h>
#include <time.h>
#include "modes.h"
#include "mdct.h"
#include "kiss_fft.h"

void clt_mdct_backward_B1_C2(const mdct_lookup *l, kiss_fft_scalar *in[2],
                             kiss_fft_scalar *OPUS_RESTRICT out[2],
                             const opus_val16 *OPUS_RESTRICT window, int overlap,
                             int shift, int stride);

static const CELTMode *ref_mode(void)
{
    static const CELTMode *m = 0;
    if (!m) m = opus_custom_mode_create(48000, 960, 0);
    return m;
}

/* trig[481], window[120], tw[480*2], bitrev[480+240+120+60] (int16), factors[4][16] (int16) */
int ref_get_tables(float *trig, float *window, float *tw, short *bitrev, short *factors)
{
    const CELTMode *m = ref_mode();
    int s, i, off = 0;
    if (!m) return -1;
    memcpy(trig, m->mdct.trig, sizeof(float) * (m->mdct.n / 4 + 1));
    memcpy(window, m->window, sizeof(float) * m->overlap);
    memcpy(tw, m->mdct.kfft[0]->twiddles, sizeof(float) * 2 * m->mdct.kfft[0]->nfft);
    for (s = 0; s <= m->mdct.maxshift; s++) {
        const kiss_fft_state *st = m->mdct.kfft[s];
        for (i = 0; i < st->nfft; i++) bitrev[off + i] = st->bitrev[i];
        off += st->nfft;
        for (i = 0; i < 16; i++) factors[16 * s + i] = st->factors[i];
    }
    return m->mdct.n;
}

int ref_overlap(void) { return ref_mode()->overlap; }

/* allocation-side tables of the static mode (static_modes_float.h:36-97): logN[21],
 * cache.index[105], cache.bits[size], cache.caps[168]; returns cache.size */
int ref_get_alloc_tables(short *logN, short *index, unsigned char *bits, unsigned char *caps)
{
    const CELTMode *m = ref_mode();
    int i;
    for (i = 0; i < m->nbEBands; i++) logN[i] = m->logN[i];
    for (i = 0; i < m->nbEBands * (m->maxLM + 2); i++) index[i] = m->cache.index[i];
    for (i = 0; i < m->cache.size; i++) bits[i] = m->cache.bits[i];
    for (i = 0; i < (m->maxLM + 1) * 2 * m->nbEBands; i++) caps[i] = m->cache.caps[i];
    return m->cache.size;
}

/* out is read-modify-write over (1920>>shift)/2 + overlap/2 floats */
void ref_imdct(float *in, float *out, int shift, int stride)
{
    const CELTMode *m = ref_mode();
    clt_mdct_backward(&m->mdct, in, out, m->window, m->overlap, shift, stride);
}

void ref_imdct_c2(float *in0, float *in1, float *out0, float *out1, int shift, int stride)
{
    const CELTMode *m = ref_mode();
    float *in[2] = { in0, in1 };
    float *out[2] = { out0, out1 };
    clt_mdct_backward_B1_C2(&m->mdct, in, out, m->window, m->overlap, shift, stride);
}

/* opus_ifft through the mode's shared plan kfft[shift] */
void ref_ifft_shared(int shift, const float *in, float *out)
{
    const CELTMode *m = ref_mode();
    opus_ifft(m->mdct.kfft[shift], (const kiss_fft_cpx *)in, (kiss_fft_cpx *)out);
}

/* Timed loop for bench.py's cpu_baseline (kind "reference"): `rows` independent
 * stride-1 rows of N2 floats, zero carry, repeated `reps` times, one thread.
 * Returns seconds. */
double ref_imdct_bench(float *in, float *out_scratch, int shift, long rows, int reps)
{
    const CELTMode *m = ref_mode();
    int N2 = (m->mdct.n >> shift) >> 1;
    int half = m->overlap >> 1;
    struct timespec a, b;
    long r;
    int k;
    clock_gettime(CLOCK_MONOTONIC, &a);
    for (k = 0; k < reps; k++)
        for (r = 0; r < rows; r++) {
            float *o = out_scratch + r * (N2 + half);
            memset(o, 0, sizeof(float) * half);
            clt_mdct_backward(&m->mdct, in + r * N2, o, m->window, m->overlap, shift, 1);
        }
    clock_gettime(CLOCK_MONOTONIC, &b);
    return (double)(b.tv_sec - a.tv_sec) + 1e-9 * (double)(b.tv_nsec - a.tv_nsec);
}
