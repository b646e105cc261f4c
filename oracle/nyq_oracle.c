/*
 * oracle/nyq_oracle.c -- CPU ORACLE.  TEST INFRASTRUCTURE ONLY.
 *
 * A plain-C scalar restatement of the reference's CELT inverse-MDCT hot path,
 * used exclusively as the parity checker (tests/, __graft_entry__.smoke(),
 * bench.py's cpu_baseline leg).  Nothing under libnyquist_amd/ or include/
 * may link, import or call this file: the product path is the HIP library
 * and fails loudly when it is missing.
 *
 * Parity status: PINNED.  tests/test_oracle.py checks this file against
 *   - the reference's bundled golden vectors test_data/ifft_{input,output}_N{60,480}.bin
 *     (copied as data to tests/golden/), and
 *   - outputs of the reference itself (oracle/_ref/libnyq_ref.so, built from
 *     /root/reference sources by oracle/Makefile; fixtures written by
 *     oracle/gen_golden.py) -- bit-exact for every size/stride/carry case.
 *
 * What is restated (paths relative to /root/reference):
 *   third_party/opus/celt/mdct.c:267-379        clt_mdct_backward
 *   third_party/opus/celt/mdct.c:258-265        clt_mdct_backward_B1_C2
 *   third_party/opus/celt/kiss_fft.c:696-747    opus_ifft (stage driver)
 *   third_party/opus/celt/kiss_fft.c:82-110     ki_bfly2
 *   third_party/opus/celt/kiss_fft.c:158-200    ki_bfly4
 *   third_party/opus/celt/kiss_fft.c:258-306    ki_bfly3
 *   third_party/opus/celt/kiss_fft.c:385-455    ki_bfly5
 *   third_party/opus/celt/kiss_fft.c:461-554    digit-reversal table, factoring, twiddles
 *   third_party/opus/celt/kiss_fft.c:749-771    test_opus_ifft (own-twiddle IFFT)
 *   third_party/opus/celt/mdct.c:70-106         trig table formula (float PI)
 *   third_party/opus/celt/modes.c:372-374       window formula
 *   third_party/opus/celt/_kiss_fft_guts.h:107-153  float complex macros
 *
 * The arithmetic keeps the reference's operation order so that, compiled
 * without FMA contraction, results are bit-identical to the reference build.
 * The code structure (plan record, stage table, row drivers) is this
 * repository's own.
 */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

#ifdef _OPENMP
#include <omp.h>
#endif

#define NQ_MDCT_N   1920   /* static mode: mdct.n, static_modes_float.h:591 */
#define NQ_OVERLAP  120    /* static_modes_float.h:579 */
#define NQ_HALF_OV  (NQ_OVERLAP / 2)
#define NQ_TW_BASE  480    /* shared twiddle table length */
#define NQ_MAX_STAGE 8

typedef struct { float re, im; } nq_cpx;

typedef struct {
    int nfft;
    int nstage;
    int radix[NQ_MAX_STAGE]; /* outermost factor first (kiss "factors[2i]")   */
    int rest[NQ_MAX_STAGE];  /* remaining length after it ("factors[2i+1]")  */
    int tw_step;             /* step into tw[] per unit fstride (1<<shift)    */
    const nq_cpx *tw;        /* tw[k] = exp(-2 pi i k / ntw)                  */
    int16_t perm[NQ_TW_BASE];/* fout[perm[i]] = fin[i]                        */
} nq_plan;

/* ---- tables --------------------------------------------------------- */

static float  g_trig[NQ_MDCT_N / 4 + 1];
static float  g_window[NQ_OVERLAP];
static nq_cpx g_tw[NQ_TW_BASE];
static nq_plan g_plan[4];          /* index = shift, nfft = 480 >> shift */
static int    g_ready = 0;

/* kiss_fft.c:498-535: peel 4s, then 2s, then odd primes up to 5 */
static int nq_factor(int n, int *radix, int *rest)
{
    int p = 4, count = 0;
    do {
        while (n % p) {
            if (p == 4) p = 2;
            else if (p == 2) p = 3;
            else p += 2;
            if (p > 32000 || p * p > n) p = n;
        }
        n /= p;
        if (p > 5) return -1;
        radix[count] = p;
        rest[count] = n;
        count++;
    } while (n > 1);
    return count;
}

/* kiss_fft.c:461-492: recursive mixed-radix digit reversal */
static void nq_digit_reverse(int out_base, int16_t *slot, int slot_step,
                             const int *radix, const int *rest, int level)
{
    int p = radix[level], m = rest[level], j;
    if (m == 1) {
        for (j = 0; j < p; j++) { *slot = (int16_t)(out_base + j); slot += slot_step; }
    } else {
        for (j = 0; j < p; j++) {
            nq_digit_reverse(out_base, slot, slot_step * p, radix, rest, level + 1);
            slot += slot_step;
            out_base += m;
        }
    }
}

static int nq_plan_build(nq_plan *pl, int nfft, const nq_cpx *tw, int tw_step)
{
    pl->nfft = nfft;
    pl->tw = tw;
    pl->tw_step = tw_step;
    pl->nstage = nq_factor(nfft, pl->radix, pl->rest);
    if (pl->nstage <= 0 || nfft > NQ_TW_BASE) return -1;
    nq_digit_reverse(0, pl->perm, 1, pl->radix, pl->rest, 0);
    return 0;
}

/* kiss_fft.c:537-554 (float branch): double-precision cexp, rounded once */
static void nq_fill_twiddles(nq_cpx *tw, int n)
{
    const double pi = 3.14159265358979323846264338327;
    int k;
    for (k = 0; k < n; k++) {
        double ph = (-2 * pi / n) * k;
        tw[k].re = (float)cos(ph);
        tw[k].im = (float)sin(ph);
    }
}

static void nq_default_tables(void)
{
    int i;
    const float PIf = 3.141592653f;                  /* mathops.h:83 */
    for (i = 0; i <= NQ_MDCT_N / 4; i++)             /* mdct.c:101-102 */
        g_trig[i] = (float)cos(2 * PIf * i / NQ_MDCT_N);
    for (i = 0; i < NQ_OVERLAP; i++) {               /* modes.c:372-374 */
        double s = sin(.5 * M_PI * (i + .5) / NQ_OVERLAP);
        g_window[i] = (float)(1.0f * sin(.5 * M_PI * s * s));
    }
    nq_fill_twiddles(g_tw, NQ_TW_BASE);
}

static void nq_build_plans(void)
{
    int s;
    for (s = 0; s < 4; s++)
        nq_plan_build(&g_plan[s], NQ_TW_BASE >> s, g_tw, 1 << s);
    g_ready = 1;
}

/* Use generated tables (formulas above). */
void nyq_oracle_init_default(void)
{
    nq_default_tables();
    nq_build_plans();
}

/* Use caller tables (e.g. the reference's static tables captured in
 * tests/golden/ref_tables.npz): trig[481], window[120], tw[480] complex. */
void nyq_oracle_init_tables(const float *trig, const float *window, const float *tw_interleaved)
{
    memcpy(g_trig, trig, sizeof g_trig);
    memcpy(g_window, window, sizeof g_window);
    memcpy(g_tw, tw_interleaved, sizeof g_tw);
    nq_build_plans();
}

static void nq_need_init(void) { if (!g_ready) nyq_oracle_init_default(); }

void nyq_oracle_get_tables(float *trig, float *window, float *tw_interleaved)
{
    nq_need_init();
    memcpy(trig, g_trig, sizeof g_trig);
    memcpy(window, g_window, sizeof g_window);
    memcpy(tw_interleaved, g_tw, sizeof g_tw);
}

/* perm table + factor list of the shared plan for `shift` (introspection for tests) */
int nyq_oracle_get_plan(int shift, int16_t *perm, int *radix, int *rest)
{
    int i;
    if (shift < 0 || shift > 3) return -1;
    nq_need_init();
    for (i = 0; i < g_plan[shift].nfft; i++) perm[i] = g_plan[shift].perm[i];
    for (i = 0; i < g_plan[shift].nstage; i++) { radix[i] = g_plan[shift].radix[i]; rest[i] = g_plan[shift].rest[i]; }
    return g_plan[shift].nstage;
}

/* ---- inverse butterflies -------------------------------------------- */
/* conj-twiddle product, _kiss_fft_guts.h:111-113 (C_MULC) */
#define NQ_MULC(d, a, w) do { (d).re = (a).re * (w).re + (a).im * (w).im; \
                              (d).im = (a).im * (w).re - (a).re * (w).im; } while (0)

static void nq_inv2(nq_cpx *f, int fs, const nq_cpx *tw, int m, int groups, int gstride)
{
    int g, j;
    for (g = 0; g < groups; g++) {
        nq_cpx *lo = f + g * gstride, *hi = lo + m;
        for (j = 0; j < m; j++) {
            nq_cpx t;
            NQ_MULC(t, hi[j], tw[j * fs]);
            hi[j].re = lo[j].re - t.re;  hi[j].im = lo[j].im - t.im;
            lo[j].re += t.re;            lo[j].im += t.im;
        }
    }
}

static void nq_inv4(nq_cpx *f, int fs, const nq_cpx *tw, int m, int groups, int gstride)
{
    int g, j;
    for (g = 0; g < groups; g++) {
        nq_cpx *p0 = f + g * gstride, *p1 = p0 + m, *p2 = p0 + 2 * m, *p3 = p0 + 3 * m;
        for (j = 0; j < m; j++) {
            nq_cpx a, b, c, sum, dif, odd;
            NQ_MULC(a, p1[j], tw[j * fs]);
            NQ_MULC(b, p2[j], tw[2 * j * fs]);
            NQ_MULC(c, p3[j], tw[3 * j * fs]);
            odd.re = p0[j].re - b.re;      odd.im = p0[j].im - b.im;
            p0[j].re += b.re;              p0[j].im += b.im;
            sum.re = a.re + c.re;          sum.im = a.im + c.im;
            dif.re = a.re - c.re;          dif.im = a.im - c.im;
            p2[j].re = p0[j].re - sum.re;  p2[j].im = p0[j].im - sum.im;
            p0[j].re += sum.re;            p0[j].im += sum.im;
            p1[j].re = odd.re - dif.im;    p1[j].im = odd.im + dif.re;
            p3[j].re = odd.re + dif.im;    p3[j].im = odd.im - dif.re;
        }
    }
}

static void nq_inv3(nq_cpx *f, int fs, const nq_cpx *tw, int m, int groups, int gstride)
{
    int g, j;
    const float neg_s60 = -tw[fs * m].im;            /* "-epi3.i" */
    for (g = 0; g < groups; g++) {
        nq_cpx *p0 = f + g * gstride, *p1 = p0 + m, *p2 = p0 + 2 * m;
        for (j = 0; j < m; j++) {
            nq_cpx a, b, sum, dif;
            NQ_MULC(a, p1[j], tw[j * fs]);
            NQ_MULC(b, p2[j], tw[2 * j * fs]);
            sum.re = a.re + b.re;   sum.im = a.im + b.im;
            dif.re = a.re - b.re;   dif.im = a.im - b.im;
            p1[j].re = p0[j].re - sum.re * .5f;
            p1[j].im = p0[j].im - sum.im * .5f;
            dif.re *= neg_s60;      dif.im *= neg_s60;
            p0[j].re += sum.re;     p0[j].im += sum.im;
            p2[j].re = p1[j].re + dif.im;
            p2[j].im = p1[j].im - dif.re;
            p1[j].re -= dif.im;
            p1[j].im += dif.re;
        }
    }
}

static void nq_inv5(nq_cpx *f, int fs, const nq_cpx *tw, int m, int groups, int gstride)
{
    int g, u;
    const nq_cpx ya = tw[fs * m], yb = tw[fs * 2 * m];
    for (g = 0; g < groups; g++) {
        nq_cpx *p0 = f + g * gstride, *p1 = p0 + m, *p2 = p0 + 2 * m, *p3 = p0 + 3 * m, *p4 = p0 + 4 * m;
        for (u = 0; u < m; u++) {
            nq_cpx z0 = p0[u], z1, z2, z3, z4, s14, d14, s23, d23, e, o;
            NQ_MULC(z1, p1[u], tw[u * fs]);
            NQ_MULC(z2, p2[u], tw[2 * u * fs]);
            NQ_MULC(z3, p3[u], tw[3 * u * fs]);
            NQ_MULC(z4, p4[u], tw[4 * u * fs]);
            s14.re = z1.re + z4.re;  s14.im = z1.im + z4.im;
            d14.re = z1.re - z4.re;  d14.im = z1.im - z4.im;
            s23.re = z2.re + z3.re;  s23.im = z2.im + z3.im;
            d23.re = z2.re - z3.re;  d23.im = z2.im - z3.im;

            p0[u].re += s14.re + s23.re;
            p0[u].im += s14.im + s23.im;

            e.re = z0.re + s14.re * ya.re + s23.re * yb.re;
            e.im = z0.im + s14.im * ya.re + s23.im * yb.re;
            o.re = -(d14.im * ya.im) - d23.im * yb.im;
            o.im = d14.re * ya.im + d23.re * yb.im;
            p1[u].re = e.re - o.re;  p1[u].im = e.im - o.im;
            p4[u].re = e.re + o.re;  p4[u].im = e.im + o.im;

            e.re = z0.re + s14.re * yb.re + s23.re * ya.re;
            e.im = z0.im + s14.im * yb.re + s23.im * ya.re;
            o.re = d14.im * yb.im - d23.im * ya.im;
            o.im = -(d14.re * yb.im) + d23.re * ya.im;
            p2[u].re = e.re + o.re;  p2[u].im = e.im + o.im;
            p3[u].re = e.re - o.re;  p3[u].im = e.im - o.im;
        }
    }
}

/* kiss_fft.c:696-747: scatter through the digit-reversal table, then run the
 * factor list innermost-first; group count of stage i = product of the outer
 * radices, twiddle step scaled by the plan's shift. */
static void nq_ifft_run(const nq_plan *pl, const nq_cpx *in, nq_cpx *out)
{
    int outer[NQ_MAX_STAGE + 1];
    int i, m;
    for (i = 0; i < pl->nfft; i++) out[pl->perm[i]] = in[i];
    outer[0] = 1;
    for (i = 0; i < pl->nstage; i++) outer[i + 1] = outer[i] * pl->radix[i];
    m = pl->rest[pl->nstage - 1];
    for (i = pl->nstage - 1; i >= 0; i--) {
        int gstride = i ? pl->rest[i - 1] : 1;
        int fs = outer[i] * pl->tw_step;
        switch (pl->radix[i]) {
        case 2: nq_inv2(out, fs, pl->tw, m, outer[i], gstride); break;
        case 4: nq_inv4(out, fs, pl->tw, m, outer[i], gstride); break;
        case 3: nq_inv3(out, fs, pl->tw, m, outer[i], gstride); break;
        case 5: nq_inv5(out, fs, pl->tw, m, outer[i], gstride); break;
        }
        m = gstride;
    }
}

/* ---- public: IFFT stage ---------------------------------------------- */

/* opus_ifft through the mode's shared plan (kfft[shift], twiddle step 1<<shift).
 * in/out: nfft interleaved complex float32, out-of-place. */
int nyq_oracle_ifft_shared(int shift, const float *in, float *out)
{
    if (shift < 0 || shift > 3 || in == out) return -1;
    nq_need_init();
    nq_ifft_run(&g_plan[shift], (const nq_cpx *)in, (nq_cpx *)out);
    return 0;
}

/* test_opus_ifft (kiss_fft.c:749-771): a fresh plan with its own nfft-long
 * twiddle table.  This is the producer shape of the bundled .bin vectors. */
int nyq_oracle_ifft_own(int nfft, const float *in, float *out)
{
    nq_plan pl;
    nq_cpx *tw;
    int rc;
    if (nfft <= 0 || nfft > NQ_TW_BASE || in == out) return -1;
    tw = (nq_cpx *)malloc(sizeof(nq_cpx) * (size_t)nfft);
    if (!tw) return -2;
    nq_fill_twiddles(tw, nfft);
    rc = nq_plan_build(&pl, nfft, tw, 1);
    if (rc == 0) nq_ifft_run(&pl, (const nq_cpx *)in, (nq_cpx *)out);
    free(tw);
    return rc;
}

/* batch of rows through nyq_oracle_ifft_own / _shared (shared=1 needs nfft = 480>>s) */
int nyq_oracle_ifft_batch(int nfft, int shared, const float *in, float *out, long batch, int nthreads)
{
    long r;
    int shift = -1, s;
    nq_plan own;
    nq_cpx *tw = NULL;
    const nq_plan *pl;
    nq_need_init();
    if (shared) {
        for (s = 0; s < 4; s++) if ((NQ_TW_BASE >> s) == nfft) shift = s;
        if (shift < 0) return -1;
        pl = &g_plan[shift];
    } else {
        if (nfft <= 0 || nfft > NQ_TW_BASE) return -1;
        tw = (nq_cpx *)malloc(sizeof(nq_cpx) * (size_t)nfft);
        if (!tw) return -2;
        nq_fill_twiddles(tw, nfft);
        if (nq_plan_build(&own, nfft, tw, 1)) { free(tw); return -1; }
        pl = &own;
    }
    (void)nthreads;
#ifdef _OPENMP
#pragma omp parallel for num_threads(nthreads > 0 ? nthreads : 1) schedule(static)
#endif
    for (r = 0; r < batch; r++)
        nq_ifft_run(pl, (const nq_cpx *)in + r * nfft, (nq_cpx *)out + r * nfft);
    free(tw);
    return 0;
}

/* ---- public: full IMDCT ---------------------------------------------- */

/* clt_mdct_backward, mdct.c:267-379, float build, static 48 kHz mode.
 * in : N2 = 960>>shift coefficients, element k at in[k*stride]
 * out: N2 + overlap/2 floats, read-modify-write (out[0..60) = carry in). */
int nyq_oracle_imdct(const float *in, float *out, int shift, int stride)
{
    float scratch[NQ_MDCT_N / 2];
    int i;
    int N, N2, N4;
    float sine;
    const float *t;
    if (shift < 0 || shift > 3 || stride < 1) return -1;
    nq_need_init();
    t = g_trig;
    N = NQ_MDCT_N >> shift;
    N2 = N >> 1;
    N4 = N >> 2;
    sine = (float)2 * 3.141592653f * (.125f) / N;      /* mdct.c:292 */

    /* pre-rotation, mdct.c:295-313 */
    {
        const float *fwd = in;
        const float *bwd = in + stride * (N2 - 1);
        float *dst = scratch;
        for (i = 0; i < N4; i++) {
            float c = t[i << shift], s = t[(N4 - i) << shift];
            float yr = -(*bwd * c) + *fwd * s;
            float yi = -(*bwd * s) - *fwd * c;
            *dst++ = yr - yi * sine;
            *dst++ = yi + yr * sine;
            fwd += 2 * stride;
            bwd -= 2 * stride;
        }
    }

    /* N/4-point unscaled inverse FFT into out + overlap/2, mdct.c:316-317 */
    nq_ifft_run(&g_plan[shift], (const nq_cpx *)scratch, (nq_cpx *)(out + NQ_HALF_OV));

    /* post-rotation from both ends, mdct.c:322-359 */
    {
        float *head = out + NQ_HALF_OV;
        float *back = out + NQ_HALF_OV + N2 - 2;
        for (i = 0; i < (N4 + 1) >> 1; i++) {
            float re = head[0], im = head[1];
            float t0 = t[i << shift], t1 = t[(N4 - i) << shift];
            float yr = re * t0 - im * t1;
            float yi = im * t0 + re * t1;
            re = back[0];
            im = back[1];
            head[0] = -(yr - yi * sine);
            back[1] = yi + yr * sine;
            t0 = t[(N4 - i - 1) << shift];
            t1 = t[(i + 1) << shift];
            yr = re * t0 - im * t1;
            yi = im * t0 + re * t1;
            back[0] = -(yr - yi * sine);
            head[1] = yi + yr * sine;
            head += 2;
            back -= 2;
        }
    }

    /* TDAC mirror, mdct.c:362-377 */
    {
        float *hi = out + NQ_OVERLAP - 1;
        float *lo = out;
        const float *wa = g_window;
        const float *wb = g_window + NQ_OVERLAP - 1;
        for (i = 0; i < NQ_OVERLAP / 2; i++) {
            float x1 = *hi, x2 = *lo;
            *lo++ = *wb * x2 - *wa * x1;
            *hi-- = *wa * x2 + *wb * x1;
            wa++;
            wb--;
        }
    }
    return 0;
}

/* clt_mdct_backward_B1_C2 (mdct.c:258-265): channel 0 then channel 1 */
int nyq_oracle_imdct_c2(const float *in0, const float *in1, float *out0, float *out1, int shift, int stride)
{
    int rc = nyq_oracle_imdct(in0, out0, shift, stride);
    if (rc) return rc;
    return nyq_oracle_imdct(in1, out1, shift, stride);
}

/* Independent rows.  in [batch][N2] contiguous (stride 1); carry [batch][60]
 * or NULL (= zeros); fin [batch][N2] = the reference's out[0..N2) after the
 * call; tail [batch][60] (may be NULL) = out[N2..N2+60), the raw values the
 * next block of the same channel receives as its carry
 * (celt_decoder_clean.c:625,641 buffer shift). */
int nyq_oracle_imdct_batch(int shift, const float *in, const float *carry,
                           float *fin, float *tail, long batch, int nthreads)
{
    long r;
    int N2;
    if (shift < 0 || shift > 3) return -1;
    nq_need_init();
    N2 = (NQ_MDCT_N >> shift) >> 1;
    (void)nthreads;
#ifdef _OPENMP
#pragma omp parallel for num_threads(nthreads > 0 ? nthreads : 1) schedule(static)
#endif
    for (r = 0; r < batch; r++) {
        float buf[NQ_MDCT_N / 2 + NQ_HALF_OV];
        if (carry) memcpy(buf, carry + r * NQ_HALF_OV, sizeof(float) * NQ_HALF_OV);
        else memset(buf, 0, sizeof(float) * NQ_HALF_OV);
        nyq_oracle_imdct(in + r * N2, buf, shift, 1);
        memcpy(fin + r * N2, buf, sizeof(float) * (size_t)N2);
        if (tail) memcpy(tail + r * NQ_HALF_OV, buf + N2, sizeof(float) * NQ_HALF_OV);
    }
    return 0;
}

/* Chained rows: rows are consecutive blocks of ONE channel; row r's carry is
 * row r-1's tail (carry0[60] or NULL seeds row 0).  pcm [batch][N2],
 * tail_out[60] = tail of the last row.  Sequential by construction. */
int nyq_oracle_imdct_chain(int shift, const float *in, const float *carry0,
                           float *pcm, float *tail_out, long batch)
{
    long r;
    int N2;
    float buf[NQ_MDCT_N / 2 + NQ_HALF_OV];
    if (shift < 0 || shift > 3) return -1;
    nq_need_init();
    N2 = (NQ_MDCT_N >> shift) >> 1;
    if (carry0) memcpy(buf, carry0, sizeof(float) * NQ_HALF_OV);
    else memset(buf, 0, sizeof(float) * NQ_HALF_OV);
    for (r = 0; r < batch; r++) {
        nyq_oracle_imdct(in + r * N2, buf, shift, 1);
        memcpy(pcm + r * N2, buf, sizeof(float) * (size_t)N2);
        memmove(buf, buf + N2, sizeof(float) * NQ_HALF_OV);
    }
    if (tail_out) memcpy(tail_out, buf, sizeof(float) * NQ_HALF_OV);
    return 0;
}

/* compute_inv_mdcts (celt_decoder_clean.c:264-312) over whole frame sequences, with the
 * decode_mem carry of celt_decoder_clean.c:622-656 emulated by one contiguous per-channel
 * buffer (frame f's out_syn = mem + f*N; the 60 floats past a block are left in place and are
 * the next block's carry).  All frames have size N = 120 << LM.
 *   freq      [nstreams][nframes][channels][N]
 *   transient [nstreams][nframes] (NULL = none): B = 2^LM interleaved short blocks, shift 3,
 *             stride B (:292-300, :301-311); otherwise one block, shift 3-LM, stride 1
 *   pcm       [nstreams][channels][nframes*N]
 *   state     [nstreams*channels][60] in/out (NULL = zeros in, discarded out)
 * The B==1&&C==2 / B==8&&C==2 fast paths of the reference are clt_mdct_backward per channel
 * (mdct.c:258-265), so one generic loop restates all three branches. */
int nyq_oracle_celt_synth(int LM, const float *freq, const unsigned char *transient, float *pcm,
                          float *state, long nstreams, long nframes, int channels, int nthreads)
{
    long sc;
    const long N = 120L << LM;
    const int B = 1 << LM;
    if (LM < 0 || LM > 3 || channels < 1) return -1;
    nq_need_init();
    (void)nthreads;
#ifdef _OPENMP
#pragma omp parallel for num_threads(nthreads > 0 ? nthreads : 1) schedule(static)
#endif
    for (sc = 0; sc < nstreams * channels; sc++) {
        const long s = sc / channels, c = sc % channels;
        long f;
        int b;
        float *mem = (float *)malloc(sizeof(float) * (size_t)(nframes * N + NQ_HALF_OV));
        if (!mem) continue;
        if (state) memcpy(mem, state + sc * NQ_HALF_OV, sizeof(float) * NQ_HALF_OV);
        else memset(mem, 0, sizeof(float) * NQ_HALF_OV);
        for (f = 0; f < nframes; f++) {
            const float *X = freq + ((s * nframes + f) * channels + c) * N;
            float *out = mem + f * N;
            if (transient && transient[s * nframes + f]) {
                for (b = 0; b < B; b++) nyq_oracle_imdct(X + b, out + 120 * b, 3, B);
            } else {
                nyq_oracle_imdct(X, out, 3 - LM, 1);
            }
        }
        memcpy(pcm + sc * nframes * N, mem, sizeof(float) * (size_t)(nframes * N));
        if (state) memcpy(state + sc * NQ_HALF_OV, mem + nframes * N, sizeof(float) * NQ_HALF_OV);
        free(mem);
    }
    return 0;
}

/* ---- post-filter and de-emphasis (the steps right after the IMDCT) ---------------------- */

/* comb_filter_const (float build: MULT16_32_Q15(a,b) = a*b).  Two variants exist in the
 * reference: the portable one, celt.c:87-110, sums left to right; on x86 (any gcc build there
 * defines __SSE__, pitch.h:40-41) celt/x86/pitch_sse.h:104-150 overrides it and adds the two
 * outer tap pairs as a partial sum first.  The reference build this oracle is pinned against
 * (oracle/_ref, x86-64) runs the SSE variant, so that association is the default here;
 * nyq_oracle_set_comb_portable(1) selects the portable order.  They differ by <= 1 ulp. */
static int g_comb_portable = 0;
void nyq_oracle_set_comb_portable(int on) { g_comb_portable = on; }

static void nq_comb_const(float *y, const float *x, int T, int N, float g10, float g11, float g12)
{
    float x0, x1, x2, x3, x4;
    int i;
    x4 = x[-T - 2];
    x3 = x[-T - 1];
    x2 = x[-T];
    x1 = x[-T + 1];
    for (i = 0; i < N; i++) {
        x0 = x[i - T + 2];
        if (g_comb_portable) {
            y[i] = x[i] + g10 * x2 + g11 * (x1 + x3) + g12 * (x0 + x4);
        } else {
            float part = g11 * (x3 + x1) + g12 * (x4 + x0);
            y[i] = (x[i] + g10 * x2) + part;
        }
        x4 = x3; x3 = x2; x2 = x1; x1 = x0;
    }
}

/* comb_filter, celt.c:114-172: cross-fade from (T0,g0,tapset0) to (T1,g1,tapset1) over the
 * first `overlap` samples with the squared window, then the constant filter.  x == y in the
 * decoder (in place on out_syn), so x[i-T] are already-filtered samples. */
void nyq_oracle_comb_filter(float *y, float *x, int T0, int T1, int N, float g0, float g1,
                            int tapset0, int tapset1, int overlap)
{
    static const float gains[3][3] = {
        {0.3066406250f, 0.2170410156f, 0.1296386719f},
        {0.4638671875f, 0.2680664062f, 0.f},
        {0.7998046875f, 0.1000976562f, 0.f}};
    float g00, g01, g02, g10, g11, g12, x0, x1, x2, x3, x4;
    const float *window;
    int i;
    nq_need_init();
    window = g_window;
    if (g0 == 0 && g1 == 0) {
        if (x != y) memmove(y, x, sizeof(float) * (size_t)N);
        return;
    }
    g00 = g0 * gains[tapset0][0];
    g01 = g0 * gains[tapset0][1];
    g02 = g0 * gains[tapset0][2];
    g10 = g1 * gains[tapset1][0];
    g11 = g1 * gains[tapset1][1];
    g12 = g1 * gains[tapset1][2];
    x1 = x[-T1 + 1];
    x2 = x[-T1];
    x3 = x[-T1 - 1];
    x4 = x[-T1 - 2];
    for (i = 0; i < overlap; i++) {
        float f;
        x0 = x[i - T1 + 2];
        f = window[i] * window[i];
        y[i] = x[i]
             + ((1.0f - f) * g00) * x[i - T0]
             + ((1.0f - f) * g01) * (x[i - T0 + 1] + x[i - T0 - 1])
             + ((1.0f - f) * g02) * (x[i - T0 + 2] + x[i - T0 - 2])
             + (f * g10) * x2
             + (f * g11) * (x1 + x3)
             + (f * g12) * (x0 + x4);
        x4 = x3; x3 = x2; x2 = x1; x1 = x0;
    }
    if (g1 == 0) {
        if (x != y) memmove(y + overlap, x + overlap, sizeof(float) * (size_t)(N - overlap));
        return;
    }
    nq_comb_const(y + i, x + i, T1, N - i, g10, g11, g12);
}

/* The post-filter and de-emphasis of celt_decode_with_ec (celt_decoder_clean.c:658-680, 723;
 * deemphasis :192-256, float build, downsample 1) over frame sequences.
 *   pcm     [nstreams*channels] rows of `pitch` floats; each row = `pre` floats of filtered
 *           history (>= 1026) followed by nframes*N IMDCT output samples; filtered in place
 *   pf_pitch/pf_gain/pf_tapset [nstreams][nframes]: the post-filter parameters decoded from
 *           each frame's bitstream (postfilter_pitch, postfilter_gain, postfilter_tapset)
 *   pf_state [nstreams][6] in/out: period_old, period, tapset_old, tapset (as floats), gain_old, gain
 *           -> layout {period_old, period, gain_old, gain, tapset_old, tapset}
 *   deemph_mem [nstreams*channels] in/out: preemph_memD
 *   out     [nstreams][nframes*N][channels] interleaved float PCM in [-1,1) (AudioData layout)
 */
int nyq_oracle_celt_post(int LM, float *pcm, long pitch, long pre, const int *pf_pitch, const float *pf_gain,
                         const int *pf_tapset, float *pf_state, float *deemph_mem, float *out,
                         long nstreams, long nframes, int channels)
{
    const long N = 120L << LM;
    const float coef0 = 0.85000610f;      /* mode->preemph[0], static_modes_float.h:582 */
    long s;
    if (LM < 0 || LM > 3 || pre < 1026) return -1;
    nq_need_init();
    for (s = 0; s < nstreams; s++) {
        int T_old = (int)pf_state[6 * s + 0], T_cur = (int)pf_state[6 * s + 1];
        float g_old = pf_state[6 * s + 2], g_cur = pf_state[6 * s + 3];
        int ts_old = (int)pf_state[6 * s + 4], ts_cur = (int)pf_state[6 * s + 5];
        long f;
        int c;
        for (f = 0; f < nframes; f++) {
            const int T_new = pf_pitch[s * nframes + f];
            const float g_new = pf_gain[s * nframes + f];
            const int ts_new = pf_tapset[s * nframes + f];
            if (T_cur < 15) T_cur = 15;            /* IMAX(.., COMBFILTER_MINPERIOD), :661-662 */
            if (T_old < 15) T_old = 15;
            for (c = 0; c < channels; c++) {
                float *syn = pcm + (s * channels + c) * pitch + pre + f * N;
                long j;
                float m = deemph_mem[s * channels + c];
                nyq_oracle_comb_filter(syn, syn, T_old, T_cur, 120, g_old, g_cur, ts_old, ts_cur, NQ_OVERLAP);
                if (LM != 0)
                    nyq_oracle_comb_filter(syn + 120, syn + 120, T_cur, T_new, (int)N - 120, g_cur, g_new, ts_cur,
                                           ts_new, NQ_OVERLAP);
                for (j = 0; j < N; j++) {              /* deemphasis :243-248 */
                    float tmp = syn[j] + m + 1e-30f;
                    m = coef0 * tmp;
                    out[((s * nframes + f) * N + j) * channels + c] = tmp * (1 / 32768.f);
                }
                deemph_mem[s * channels + c] = m;
            }
            T_old = T_cur; g_old = g_cur; ts_old = ts_cur;      /* :672-677 */
            T_cur = T_new; g_cur = g_new; ts_cur = ts_new;
            if (LM != 0) { T_old = T_cur; g_old = g_cur; ts_old = ts_cur; }   /* :678-683 */
        }
        pf_state[6 * s + 0] = (float)T_old; pf_state[6 * s + 1] = (float)T_cur;
        pf_state[6 * s + 2] = g_old;        pf_state[6 * s + 3] = g_cur;
        pf_state[6 * s + 4] = (float)ts_old; pf_state[6 * s + 5] = (float)ts_cur;
    }
    return 0;
}

/* ---- Vorbis inverse MDCT ---------------------------------------------- */
/* libvorbis mdct_backward (third_party/libvorbis/src/mdct.c:397-491) evaluated from its closed form in
 * double precision:  out[i] = sum_k in[k] cos(2 pi / n (i + 1/2 + n/4)(k + 1/2)),  i < n, k < n/2.
 * O(n^2): a known-answer restatement, pinned to outputs of the compiled libvorbis file
 * (oracle/_ref/libvorbis_ref.so, fixtures tests/golden/ref_vorbis.npz) to <= 1e-6 relative RMS --
 * libvorbis' own factorised float arithmetic is not reproduced bit for bit. */
int nyq_oracle_vorbis_imdct(int n, const float *in, float *out, long rows)
{
    long r;
    int i, k;
    double *ctab;
    if (n < 16 || (n & (n - 1))) return -1;
    /* cos(2 pi m / (4n)) for m < 4n: (2i + 1 + n/2)(2k + 1) is an integer, reduce it mod 4n */
    ctab = (double *)malloc(sizeof(double) * 4 * (size_t)n);
    if (!ctab) return -2;
    for (i = 0; i < 4 * n; i++) ctab[i] = cos(2.0 * M_PI * i / (4.0 * n));
    for (r = 0; r < rows; r++) {
        const float *x = in + r * (n / 2);
        float *y = out + r * n;
        for (i = 0; i < n; i++) {
            double acc = 0;
            const long a = 2L * i + 1 + n / 2;
            for (k = 0; k < n / 2; k++) acc += x[k] * ctab[(a * (2L * k + 1)) % (4L * n)];
            y[i] = (float)acc;
        }
    }
    free(ctab);
    return 0;
}

int nyq_oracle_max_threads(void)
{
#ifdef _OPENMP
    return omp_get_max_threads();
#else
    return 1;
#endif
}
