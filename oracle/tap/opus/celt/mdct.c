/*
 * oracle/tap/opus/celt/mdct.c -- CAPTURE TAP.  TEST INFRASTRUCTURE ONLY.
 *
 * The reference's unity build (src/OpusDependencies.c:88) does `#include "opus/celt/mdct.c"`.
 * oracle/Makefile puts oracle/tap/ first on the include path, so that line finds THIS file,
 * which includes the reference's real mdct.c (read in place, never copied) under renamed entry
 * points and re-exports the original names as recording wrappers.  The callers
 * (celt_decoder_clean.c:264-312, same translation unit) therefore run the real code and every
 * call is logged.  The reference tree is not modified.
 *
 * What is recorded per clt_mdct_backward call: the pointers and (shift, stride), the N2 input
 * coefficients as read through `stride`, and out[0 .. N2+overlap/2) right after the call.
 * oracle/ref_capture.cpp turns that log into per-frame fixtures.
 */
#define clt_mdct_backward        nyqref_clt_mdct_backward
#define clt_mdct_backward_B1_C2  nyqref_clt_mdct_backward_B1_C2
#include NYQ_REAL_MDCT_C          /* "/root/reference/third_party/opus/celt/mdct.c", from the Makefile */
#undef clt_mdct_backward
#undef clt_mdct_backward_B1_C2

#include <stdlib.h>
#include <string.h>

typedef struct {
    const float *in_ptr;
    float *out_ptr;
    int shift, stride, n2;
    float *in_copy;    /* n2 coefficients, de-strided */
    float *out_copy;   /* n2 + overlap/2 floats after the call */
} nyq_tap_call;

static nyq_tap_call *g_tap_calls = 0;
static long g_tap_count = 0, g_tap_cap = 0;
static long g_tap_limit = 0;   /* 0 = recording off */

void nyq_tap_start(long max_calls) { g_tap_limit = max_calls; g_tap_count = 0; }
long nyq_tap_count(void) { return g_tap_count; }
const nyq_tap_call *nyq_tap_get(long i) { return (i >= 0 && i < g_tap_count) ? &g_tap_calls[i] : 0; }

static void nyq_tap_record(const float *in, float *out, int shift, int stride, int n, int overlap)
{
    nyq_tap_call *c;
    int k, n2 = (n >> shift) >> 1;
    if (g_tap_count >= g_tap_limit) return;
    if (g_tap_count == g_tap_cap) {
        g_tap_cap = g_tap_cap ? 2 * g_tap_cap : 1024;
        g_tap_calls = (nyq_tap_call *)realloc(g_tap_calls, sizeof(nyq_tap_call) * (size_t)g_tap_cap);
    }
    c = &g_tap_calls[g_tap_count++];
    c->in_ptr = in; c->out_ptr = out; c->shift = shift; c->stride = stride; c->n2 = n2;
    c->in_copy = (float *)malloc(sizeof(float) * (size_t)n2);
    c->out_copy = (float *)malloc(sizeof(float) * (size_t)(n2 + overlap / 2));
    for (k = 0; k < n2; k++) c->in_copy[k] = in[(size_t)k * stride];
    memcpy(c->out_copy, out, sizeof(float) * (size_t)(n2 + overlap / 2));
}

void clt_mdct_backward(const mdct_lookup *l, kiss_fft_scalar *in, kiss_fft_scalar *OPUS_RESTRICT out,
                       const opus_val16 *OPUS_RESTRICT window, int overlap, int shift, int stride)
{
    /* the input is consumed before the output is written, but copy it first anyway */
    float keep[960];
    int k, n2 = (l->n >> shift) >> 1;
    for (k = 0; k < n2; k++) keep[k] = in[(size_t)k * stride];
    nyqref_clt_mdct_backward(l, in, out, window, overlap, shift, stride);
    if (g_tap_limit) {
        nyq_tap_record(in, out, shift, stride, l->n, overlap);
        if (g_tap_count) memcpy(g_tap_calls[g_tap_count - 1].in_copy, keep, sizeof(float) * (size_t)n2);
    }
}

void clt_mdct_backward_B1_C2(const mdct_lookup *l, kiss_fft_scalar *in[2], kiss_fft_scalar *OPUS_RESTRICT out[2],
                             const opus_val16 *OPUS_RESTRICT window, int overlap, int shift, int stride)
{
    /* same order as the reference wrapper (mdct.c:258-265): channel 0, then channel 1 */
    clt_mdct_backward(l, in[0], out[0], window, overlap, shift, stride);
    clt_mdct_backward(l, in[1], out[1], window, overlap, shift, stride);
}
