/*
 * oracle/tap/opus/celt/celt.c -- CAPTURE TAP for the post-filter.  TEST INFRASTRUCTURE ONLY.
 * Same mechanism as oracle/tap/opus/celt/mdct.c: the unity build's `#include "opus/celt/celt.c"`
 * (src/OpusDependencies.c) finds this file first; it includes the reference's real celt.c with
 * comb_filter renamed and re-exports comb_filter as a recording wrapper.  The decoder's call
 * sites (celt_decoder_clean.c:663-669, a later file of the same translation unit) are logged.
 */
#define comb_filter nyqref_comb_filter
#include NYQ_REAL_CELT_C          /* "/root/reference/third_party/opus/celt/celt.c" */
#undef comb_filter

#include <stdlib.h>
#include <string.h>

#define NYQ_TAP_HIST 1088         /* DECODE_BUFFER_SIZE - 960: filtered history in front of out_syn */

typedef struct {
    float *y;
    int T0, T1, N, tapset0, tapset1;
    float g0, g1;
    float *hist;                   /* y[-1088 .. 0) before the call (first call of a frame only) */
} nyq_comb_call;

static nyq_comb_call *g_comb_calls = 0;
static long g_comb_count = 0, g_comb_cap = 0, g_comb_limit = 0;

void nyq_comb_tap_start(long max_calls) { g_comb_limit = max_calls; g_comb_count = 0; }
long nyq_comb_tap_count(void) { return g_comb_count; }
const nyq_comb_call *nyq_comb_tap_get(long i) { return (i >= 0 && i < g_comb_count) ? &g_comb_calls[i] : 0; }

void comb_filter(opus_val32 *y, opus_val32 *x, int T0, int T1, int N, opus_val16 g0, opus_val16 g1,
                 int tapset0, int tapset1, const opus_val16 *window, int overlap)
{
    if (g_comb_count < g_comb_limit) {
        nyq_comb_call *c;
        if (g_comb_count == g_comb_cap) {
            g_comb_cap = g_comb_cap ? 2 * g_comb_cap : 1024;
            g_comb_calls = (nyq_comb_call *)realloc(g_comb_calls, sizeof(nyq_comb_call) * (size_t)g_comb_cap);
        }
        c = &g_comb_calls[g_comb_count++];
        c->y = y; c->T0 = T0; c->T1 = T1; c->N = N; c->g0 = g0; c->g1 = g1;
        c->tapset0 = tapset0; c->tapset1 = tapset1;
        c->hist = 0;
        if (N == 120 && x == y) {          /* first call of a frame (N = shortMdctSize), decoder side */
            c->hist = (float *)malloc(sizeof(float) * NYQ_TAP_HIST);
            memcpy(c->hist, y - NYQ_TAP_HIST, sizeof(float) * NYQ_TAP_HIST);
        }
    }
    nyqref_comb_filter(y, x, T0, T1, N, g0, g1, tapset0, tapset1, window, overlap);
}
