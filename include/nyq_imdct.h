/*
 * nyq_imdct.h -- C ABI of libnyq_imdct.so: the MI355X (gfx950) batched CELT inverse MDCT.
 *
 * This is the drop-in boundary for ONE hot path of dafx/libnyquist: the path the
 * reference itself swaps out at third_party/opus/celt/mdct.c:219-254 onto the
 * extern "C" symbols of cuda/mdct_cuda.hpp:79-103.  Two families are exported:
 *
 *   1. the reference's own operator names (processMDCTCuda, processMDCTCudaB1C2,
 *      cleanupCudaBuffers, printCudaVersion) with identical signatures and pointer
 *      ownership, so the reference's USE_CUDA call sites link against this library
 *      unchanged;
 *   2. the batched entry points the reference lacks (one call = many
 *      clt_mdct_backward rows), which is what makes a GPU worthwhile.
 *
 * Plain pointers and sizes only; float32 little-endian everywhere.  Every function
 * that can fail returns an int status (0 = NYQ_OK); nyq_last_error() gives the text.
 * There is NO CPU fallback: without a usable HIP device nyq_ctx_create() fails with
 * NYQ_ERR_NO_DEVICE and the reference-named void shims abort() after printing why.
 *
 * Row semantics (one row == one clt_mdct_backward(l, in, out, window, 120, shift, 1),
 * third_party/opus/celt/mdct.c:267-379, static 48 kHz mode, mdct.n = 1920):
 *   N2 = 960 >> shift            coefficients in, finished samples out  (shift 0..3)
 *   in   [N2]                    frequency coefficients, stride 1
 *   carry[60]                    what the reference finds in out[0..60) on entry
 *                                (= raw tail of the previous block of that channel,
 *                                 celt_decoder_clean.c:625,641); NULL means zeros
 *   fin  [N2]                    the reference's out[0..N2) after the call
 *   tail [60]                    the reference's out[N2..N2+60) after the call
 */
#ifndef NYQ_IMDCT_H
#define NYQ_IMDCT_H

#include <stddef.h>

#ifdef __cplusplus
extern "C" {
#endif

#define NYQ_OK             0
#define NYQ_ERR_INVALID   (-1)  /* bad argument (NULL, shift/nfft out of range, misaligned pointer) */
#define NYQ_ERR_NO_DEVICE (-2)  /* no HIP device / device index out of range */
#define NYQ_ERR_HIP       (-3)  /* a HIP runtime call failed; see nyq_last_error() */
#define NYQ_ERR_ALLOC     (-4)  /* host or device allocation failed */

#define NYQ_MDCT_N   1920       /* static_modes_float.h:591 */
#define NYQ_OVERLAP  120        /* static_modes_float.h:579 */
#define NYQ_HALF_OV  60

typedef struct nyq_ctx nyq_ctx; /* one per (device, host thread); not thread-safe, replaces the
                                   reference's unsynchronised global state map, mdct_cuda.cu:558-559 */

/* ---- context ----------------------------------------------------------- */
/* number of HIP devices this process sees (0 if there is none or the runtime fails); valid device indices
 * are 0 .. count-1.  (The reference's offload has no such call: it uses device 0 implicitly, mdct_cuda.cu:318.) */
int  nyq_device_count(void);
int  nyq_ctx_create(nyq_ctx **out, int device);
void nyq_ctx_destroy(nyq_ctx *ctx);
/* text of the last failure on ctx (ctx == NULL: last failure of ctx-less calls) */
const char *nyq_last_error(const nyq_ctx *ctx);
/* Run on a caller-owned hipStream_t, taken literally: NULL is HIP's default (null) stream.
 * A new context runs on a private non-blocking stream; nyq_ctx_reset_stream() returns to it. */
int  nyq_ctx_set_stream(nyq_ctx *ctx, void *hip_stream);
int  nyq_ctx_reset_stream(nyq_ctx *ctx);
void *nyq_ctx_get_stream(nyq_ctx *ctx);
int  nyq_ctx_synchronize(nyq_ctx *ctx);
/* Replace the built-in tables by the caller's (host pointers, copied): trig[481] =
 * mode->mdct.trig (static_modes_float.h:477), window[120] = mode->window (:9-34).
 * The built-in tables are regenerated from the reference's formulas
 * (mdct.c:101-102 with mathops.h:83's float PI; modes.c:372-374). */
int  nyq_ctx_set_tables(nyq_ctx *ctx, const float *trig481, const float *window120);
int  nyq_ctx_get_tables(nyq_ctx *ctx, float *trig481, float *window120);
/* device properties the benchmark reports: compute units, waves resident per CU for `shift` */
int  nyq_ctx_device_info(nyq_ctx *ctx, int *compute_units, char *name, size_t name_len);

/* Per-context options (replace the process-environment switches of earlier builds: nothing in this library reads
 * the environment at launch time).  Set between calls, from the thread that owns the context.
 *   NYQ_OPT_BLOCKS_PER_CU   workgroups per compute unit of the persistent row kernels; 0 = the built-in choice
 *                           (a profiling knob: the occupancy caches of the context are dropped when it changes)
 *   NYQ_OPT_POST_FORM       kernel form of nyq_celt_post_dev: NYQ_POST_FORM_PIPELINE (default, the only form in the
 *                           product build), NYQ_POST_FORM_WAVE_PER_CHANNEL, NYQ_POST_FORM_WAVE_PER_PAIR
 *   NYQ_OPT_CHAIN_FUSED     nyq_celt_chain_dev, where nyq_celt_chain_fused_supported(): NYQ_CHAIN_ONE_LAUNCH (default: the
 *                           post-filter's workgroup plus a transform wave working in place in its frame regions),
 *                           NYQ_CHAIN_TWO_KERNELS (synthesis + post-filter through d_pcm, the form of every other shape),
 *                           NYQ_CHAIN_FUSED_R2 (round 2's fused kernel: A/B build only, 3x slower)
 *   NYQ_OPT_CHAIN_WINDOW    nyq_celt_chain_dev: frames per time window of the two-kernel chain, rounded up to the frame
 *                           size's chain length (16 / 32 / 32 / 64 frames for LM 3 / 2 / 1 / 0) so that the result is
 *                           bit-identical to one window; 0 = the built-in choice
 *   NYQ_OPT_CHAIN_OVERLAP   with windows: 1 = the post-filter of window k on a second stream beside the synthesis of
 *                           window k + 1 (an A/B form like the two below: measured, not faster -- DESIGN.md 4.8)
 * The alternative forms (POST_FORM != pipeline, CHAIN_FUSED = FUSED_R2, CHAIN_OVERLAP = 1) are measured-and-rejected designs kept for A/B
 * runs: they exist only in the tools' build of this library (-DNYQ_AB_FORMS, tools/libnyq_imdct_ab.so);
 * the product build answers NYQ_ERR_INVALID for them.  nyq_ab_forms_built() tells which build this is. *   NYQ_OPT_HOST_WINDOW     nyq_celt_frames_to_pcm* / nyq_celt_symbols_*_to_pcm_mapped on few long streams: frames per time window
 *                           (a multiple of 64; upload / kernels / download of consecutive windows overlap, bit-identical); 0 = the
 *                           built-in choice (about 24 MB of input per window)
 */
#define NYQ_OPT_BLOCKS_PER_CU 1
#define NYQ_OPT_POST_FORM     2
#define NYQ_OPT_CHAIN_FUSED   3
#define NYQ_OPT_CHAIN_WINDOW  4
#define NYQ_OPT_CHAIN_OVERLAP 5
#define NYQ_OPT_HOST_WINDOW   6
#define NYQ_CHAIN_TWO_KERNELS 0
#define NYQ_CHAIN_ONE_LAUNCH  1
#define NYQ_CHAIN_FUSED_R2    2
#define NYQ_POST_FORM_PIPELINE         0
#define NYQ_POST_FORM_WAVE_PER_CHANNEL 1
#define NYQ_POST_FORM_WAVE_PER_PAIR    2
int  nyq_ctx_set_option(nyq_ctx *ctx, int option, long value);
int  nyq_ctx_get_option(nyq_ctx *ctx, int option, long *value);
int  nyq_ab_forms_built(void);

/* ---- device-resident batched operators (asynchronous on the context stream) ---- */
/* All d_* pointers are device memory, 16-byte aligned, rows contiguous. No allocation,
 * no synchronisation inside: safe to capture in a hipGraph. */

/* opus_ifft (kiss_fft.c:696-747): [batch][nfft] interleaved complex in/out, natural
 * order, unscaled inverse; nfft in {480,240,120,60}.  Out of place. */
int nyq_ifft_batch_dev(nyq_ctx *ctx, int nfft, const float *d_in, float *d_out, size_t batch);

/* clt_mdct_backward (mdct.c:267-379) on `batch` independent rows.
 * d_in [batch][N2], d_carry [batch][60] or NULL, d_fin [batch][N2], d_tail [batch][60] or NULL. */
int nyq_imdct_batch_dev(nyq_ctx *ctx, int shift, const float *d_in, const float *d_carry,
                        float *d_fin, float *d_tail, size_t batch);

/* `nchains` channels of `len` consecutive same-size blocks each (rows of chain c are
 * c*len .. c*len+len-1): block r's carry is block r-1's tail, as the decoder's buffer
 * shift arranges (celt_decoder_clean.c:625,641).  d_carry0 [nchains][60] or NULL seeds
 * row 0 of each chain; d_pcm [nchains*len][N2]; d_tail_out [nchains][60] or NULL gets
 * each chain's final tail; d_work [nchains][len+1][60] is caller-provided scratch.
 * (Same machinery as nyq_celt_synth_dev with one channel and no transient frames.) */
int nyq_imdct_chain_dev(nyq_ctx *ctx, int shift, const float *d_in, const float *d_carry0,
                        float *d_pcm, float *d_tail_out, float *d_work,
                        size_t nchains, size_t len);

/* compute_inv_mdcts (celt_decoder_clean.c:264-312) plus the decode_mem carry of :622-656, for
 * whole frame sequences: `nstreams` independent decoders x `nframes` frames x `channels`
 * channels, all of frame size N = 120 << LM (LM 0..3; Opus 20 ms frames are LM 3).
 *   d_freq      [nstreams][nframes][channels][N]  freq[] exactly as denormalise_bands leaves it
 *                                                 (channel c at +c*N); in a transient frame the
 *                                                 B = 2^LM short blocks are interleaved,
 *                                                 coefficient k of block b at [b + B*k] (:292-300)
 *   d_transient [nstreams][nframes] bytes         non-zero = transient (shortBlocks) frame;
 *                                                 NULL = no transient frames
 *   d_pcm       [nstreams][channels][nframes*N]   time-contiguous per channel: what out_syn /
 *                                                 decode_mem holds before comb_filter (:663)
 *   d_state     [nstreams*channels][60] or NULL   in: overlap carry before frame 0 (zeros after a
 *                                                 reset, :846-859); out: carry after the last frame
 *   d_work      nyq_celt_synth_work_floats() floats of caller-provided scratch
 * Every raw IMDCT runs in parallel; consecutive long frames chain their carry inside a wave,
 * all other block boundaries are completed by one tiny fix-up launch.  Asynchronous. */
size_t nyq_celt_synth_work_floats(size_t nstreams, size_t nframes, int channels);
int nyq_celt_synth_dev(nyq_ctx *ctx, int LM, const float *d_freq, const unsigned char *d_transient,
                       float *d_pcm, float *d_state, float *d_work,
                       size_t nstreams, size_t nframes, int channels);

/* What celt_decode_with_ec does after the IMDCT (celt_decoder_clean.c:658-683, 723): the pitch
 * post-filter (comb_filter, celt.c:114-172) and deemphasis() (:192-256) with its 1/32768 scaling
 * and channel interleave -- i.e. from out_syn to the interleaved [-1,1) float samples that end up
 * in nqr::AudioData::samples.  Recursive along time: one wavefront per (stream, channel).
 *   d_pcm        [nstreams*channels][nframes*N]  output of nyq_celt_synth_dev (read only)
 *   d_pf_pitch, d_pf_gain, d_pf_tapset [nstreams][nframes]  post-filter parameters decoded from each
 *                frame (postfilter_pitch, postfilter_gain, postfilter_tapset; gain 0 = filter off)
 *   d_pf_state_in / d_pf_state_out [nstreams][6] or NULL  {period_old, period, gain_old, gain,
 *                tapset_old, tapset} before / after; must not alias each other
 *   d_hist       [nstreams*channels][1088] or NULL  filtered history before frame 0, in/out
 *   d_deemph     [nstreams*channels] or NULL        preemph_memD, in/out
 *   d_out        [nstreams][nframes*N][channels]    interleaved float PCM
 * NULL state pointers mean a freshly reset decoder (zeros) and discard the final state. */
int nyq_celt_post_dev(nyq_ctx *ctx, int LM, const float *d_pcm, const int *d_pf_pitch,
                      const float *d_pf_gain, const int *d_pf_tapset, const float *d_pf_state_in,
                      float *d_pf_state_out, float *d_hist, float *d_deemph, float *d_out,
                      size_t nstreams, size_t nframes, int channels);

/* Round size of nyq_celt_post_dev in (stream, channel) chains: the kernel keeps 8 chains per compute unit resident for the
 * whole launch (2048 on an MI355X), so a call runs in ceil(nstreams * channels / this) rounds of roughly equal duration --
 * 1024 stereo streams take one round, 1280 take two.  Batch callers should size calls in multiples of it (the host-buffer
 * entry points below cut their pieces that way). */
size_t nyq_celt_post_round_chains(nyq_ctx *ctx);

/* The two stages above as ONE operator: freq[] -> interleaved PCM, everything celt_decode_with_ec does after
 * denormalise_bands (celt_decoder_clean.c:620-723).  For 20 ms stereo frames (LM 3, channels 2:
 * nyq_celt_chain_fused_supported) it is a single launch in which the time-domain frame never leaves the CU
 * (3840 B in + 3840 B out per channel-frame instead of twice that); every other shape runs the two kernels
 * above through d_pcm ([nstreams*channels][nframes*N]) and d_work (nyq_celt_synth_work_floats), which the
 * fused shape ignores (NULL allowed).  State pointers as in the two stages: d_overlap [nstreams*channels][60],
 * d_hist [nstreams*channels][1088], d_deemph [nstreams*channels], d_pf_state_in/out [nstreams][6]; NULL = reset
 * decoder / discard.  Asynchronous, no allocation. */
int nyq_celt_chain_fused_supported(int LM, int channels);
int nyq_celt_chain_dev(nyq_ctx *ctx, int LM, const float *d_freq, const unsigned char *d_transient,
                       const int *d_pf_pitch, const float *d_pf_gain, const int *d_pf_tapset,
                       const float *d_pf_state_in, float *d_pf_state_out, float *d_overlap, float *d_hist,
                       float *d_deemph, float *d_out, float *d_pcm, float *d_work,
                       size_t nstreams, size_t nframes, int channels);

/* ---- the samples straight into the FILE's layout ----
 * What opus_multistream_decode_native does with every elementary stream's PCM (opus_copy_channel_out_float,
 * opus_multistream_decoder.c:305-331, call sites :237-290), what opusfile does with the stream's ends (skip pre_skip
 * samples, stop at the last granule position) and the header gain (OPUS_SET_GAIN, opus_decoder_clean.c:700-712) --
 * done by the kernels' own store phase instead of a per-sample pass on the host: a stream with a destination record
 * writes sample ts of its channel k to  base[(ts - first) * cstride + coff[k]]  (times gain) for first <= ts < last and
 * nowhere else.  `base` is device memory (nyq_device_alloc); records are per elementary stream (mono or stereo). */
typedef struct nyq_out_desc {
    float *base;            /* device: where stream sample `first`, destination channel slot 0 goes; NULL = dense output */
    long long first, last;  /* stream samples [first, last) are written (pre-skip / end trim), the rest is dropped */
    long long t0;           /* stream sample index of the call's first sample (a time slice of a longer stream) */
    int cstride;            /* floats from one sample to the next in the destination = the file's channel count */
    int coff[2];            /* destination channel slot of the stream's channel 0 / 1; -1 = not written */
    float gain;             /* multiplied in; 1 = none */
} nyq_out_desc;
/* nyq_celt_chain_dev with per-stream destinations: d_desc [nstreams] (device memory) or NULL; channels 1 or 2.  d_out
 * (the dense output) may be NULL when every record has a base. */
int nyq_celt_chain_mapped_dev(nyq_ctx *ctx, int LM, const float *d_freq, const unsigned char *d_transient,
                              const int *d_pf_pitch, const float *d_pf_gain, const int *d_pf_tapset,
                              const float *d_pf_state_in, float *d_pf_state_out, float *d_overlap, float *d_hist,
                              float *d_deemph, float *d_out, const nyq_out_desc *d_desc, float *d_pcm, float *d_work,
                              size_t nstreams, size_t nframes, int channels);
/* device memory for such destinations (on the context's device; synchronous calls) */
void *nyq_device_alloc(nyq_ctx *ctx, size_t bytes);              /* NULL on failure */
void nyq_device_free(nyq_ctx *ctx, void *d_p);
int nyq_device_zero(nyq_ctx *ctx, void *d_p, size_t bytes);      /* silent channels (mapping 255) stay zero */
int nyq_device_download(nyq_ctx *ctx, void *host_dst, const void *d_src, size_t bytes);
/* destination channel src_slot repeated in dst_slot (a mapping may name one decoded channel twice) */
int nyq_device_dup_channel(nyq_ctx *ctx, float *d_base, int cstride, int src_slot, int dst_slot, size_t nsamples);

/* ---- band shapes on the device: the host sends SYMBOLS, not coefficients (round 4) ----
 * What celt_decode_with_ec computes between the range decoder and denormalise_bands' output -- quant_all_bands'
 * arithmetic (bands.c:1355-1518: pulse vectors from their codewords cwrs.c:decode_pulses, unit-norm coefficients, the
 * spreading rotation vq.c:65-111, collapse masks, folding and noise filling, Haar / Hadamard resolution changes, mid / side
 * merging) and denormalise_bands (bands.c:192-256) -- needs no bit of the stream once the SYMBOLS are known.  The host's
 * entropy stage therefore stops at the symbols (the codeword of every pulse vector, the leaves of every band's split tree,
 * a short program of vector operations, the band gains) and nyq_celt_shape_dev builds freq[] from them on the device, one
 * wavefront per frame.  20 ms frames (LM 3), mono or stereo.  One frame = one record, at most nyq_celt_symbol_bytes(channels)
 * bytes (the slot of the fixed-stride forms), laid out compactly:
 *     nyq_sym_head | float log_gain[42] (energy + mean per band in log2 units, channel-major: the device raises 2 to it) | nyq_sym_op ops[nops] | nyq_sym_vec vecs[nvecs] |
 *     nyq_sym_leaf leaves[nleaves] (+ float level[42] with NYQ_SYM_ANTI_COLLAPSE)
 *     --  or, with NYQ_SYM_HOST_FREQ:  nyq_sym_head | float freq[channels * 960]
 * A record of zeros is a silent frame.  Offsets inside a frame are in floats of X (channel c at c * 960).  head.channels is
 * what the PACKET codes (the TOC's stereo flag); where it differs from the stream's `channels` the device duplicates the one
 * coded channel or mixes the two down, as celt_decoder_clean.c:648-652 does. */
#define NYQ_SYM_HOST_FREQ 1          /* flags: the body holds freq[] computed on the host (overlong leaf lists, frames that are not 20 ms) */
#define NYQ_SYM_ANTI_COLLAPSE 2      /* flags: float level[42] (channel-major) follows the leaves: short blocks of a transient frame that
                                        received nothing are filled with noise at that level, then the band is renormalised (bands.c:258-351) */
#define NYQ_SYM_MAX_OPS   113
#define NYQ_SYM_MAX_VECS  44
typedef struct nyq_sym_head {
    unsigned int seed;               /* state of the noise generator when the frame's band loop starts (bands.c:61-64) */
    unsigned short nleaves, nvecs, nops;
    unsigned char flags, spread, start, end, channels, lm;
    unsigned int reserved[4];
} nyq_sym_head;                      /* 32 bytes */
typedef struct nyq_sym_leaf {        /* a leaf of a band's split tree (quant_partition's recursion ends, bands.c:1005-1052) */
    short off, n, k;                 /* offset inside its vector, bins, pulses */
    unsigned char blocks, kind;      /* interleaved short blocks; 0: a pulse vector, 1: no pulses (zeros / noise / folded copy,
                                        decided on the device from the collapse masks of the bands below) */
    float gain;
    short fold_off;                  /* offset of its source inside the band's fold source, -1: none (noise) */
    unsigned char shift, pad;        /* where its blocks sit in the band's collapse mask */
    unsigned int index;              /* codeword of the pulse vector (decode_pulses' ec_dec_uint) */
    short abs, pad2;                 /* offset in X */
    unsigned short img[8];           /* its fill mask = OR of img[i] over the set bits i of the vector's initial fill mask */
} nyq_sym_leaf;                      /* 40 bytes */
typedef struct nyq_sym_vec {         /* one vector (one channel, or mid / side) of one band */
    short x, n, fold, out, nb_tree, leaf0, leaf1;
    unsigned char sel, recombine, time_divide, b_tree, b_in;
    unsigned char band;
    unsigned char cm_ch;             /* its collapse mask is ORed into the band's mask of channel 1: first, 2: second, 3: both */
    unsigned char fill_mode;         /* initial fill mask: 0 both channels' masks of bands [fill_lo, fill_hi) ORed, 1 the first
                                        channel's, 2 the second's, 3 all blocks */
    unsigned char fill_lo, fill_hi;
} nyq_sym_vec;                       /* 24 bytes */
typedef struct nyq_sym_op {          /* kinds: 0 vector a | 1 single X[a] = f0 (copy to fold memory b, channel n, if b >= 0; band's masks = 1) |
                                        2 two-bin stereo pair a, b (mid f0, side f1, n = sign | swap << 1) | 3 merge a, b over n bins (mid f0) |
                                        4 negate a over n | 5 average the two channels' fold memories over a bins */
    unsigned char kind, band;
    short a, b, n;
    float f0, f1;
} nyq_sym_op;                        /* 16 bytes */
size_t nyq_celt_symbol_bytes(int channels);                 /* the slot of a 20 ms frame */
size_t nyq_celt_symbol_bytes_lm(int channels, int LM);      /* ... of a frame of 120 << LM samples */
/* d_sym [nstreams][sstride frames][record] -> d_freq [nstreams][nframes][channels][960] (dense); sstride = frames per stream
 * in d_sym (0 = nframes).  Asynchronous on the context stream. */
int nyq_celt_shape_dev(nyq_ctx *ctx, const void *d_sym, float *d_freq, size_t nstreams, size_t nframes, int channels,
                       size_t sstride);
/* the same for frames of 120 << LM samples: slots of nyq_celt_symbol_bytes_lm, d_freq [nstreams][nframes][channels][120 << LM] */
int nyq_celt_shape_lm_dev(nyq_ctx *ctx, int LM, const void *d_sym, float *d_freq, size_t nstreams, size_t nframes, int channels,
                          size_t sstride);

/* ---- The entropy stage itself on the device: frames' BYTES -> symbol records --------------------------------------------------
 * Replaces, per frame, what the reference does between ec_dec_init and the end of quant_all_bands' bit reading
 * (celt_decoder_clean.c:462-640, quant_bands.c:427-540, rate.c:247-638, bands.c:661-1518 as far as symbols go).  Nothing a CELT
 * frame's symbols decide depends on another frame, so every frame is one GPU lane (csrc/nyq_entropy_core.hpp); what does cross
 * frames -- band energies, the noise seed, anti-collapse levels -- is folded in by a second kernel, a wave per stream.
 * The records are SPREAD: head.reserved[0] = ops offset | vecs offset << 16, reserved[1] = leaves offset | level offset << 16
 * (bytes from the record's start; zero words = the compact form above); nyq_celt_shape_* read both forms.
 * d_tables: nyq_celt_entropy_tables_bytes() bytes, filled by the host library (nyqh_entropy_tables) and uploaded by the caller.
 * d_payload (payload_bytes long) / d_desc [nstreams][nframes]: every frame's bytes and where they are (a descriptor that points
 * outside the payload is an empty frame); d_sym [nstreams][nframes][slot of
 * nyq_celt_symbol_bytes_lm(channels, LM)]; d_info [nstreams][nframes]; d_energy: scratch, NYQ_ENT_ENERGY_BYTES per frame;
 * d_state [nstreams]: the streams' energies and final range, read unless `fresh` (streams that start here), always written.
 * A frame whose lists outgrow the slot comes back with NYQ_ENT_TOO_LARGE and a silent record (never with the full slot below).
 * Asynchronous on the context stream. */
typedef struct nyq_ent_desc {
    unsigned int offset;             /* of the frame's first byte in d_payload */
    unsigned short len;              /* bytes (0..1275) */
    unsigned char channels;          /* what the packet codes (TOC stereo flag): 1 or 2 */
    unsigned char start, end;        /* coded bands [start, end): 0 and 13 / 17 / 19 / 21 by the TOC's bandwidth */
    unsigned char pad[3];
} nyq_ent_desc;                      /* 12 bytes */
#define NYQ_ENT_TRANSIENT 1
#define NYQ_ENT_SILENCE 2
#define NYQ_ENT_INTRA 4
#define NYQ_ENT_ANTI_COLLAPSE 8
#define NYQ_ENT_ERROR 16             /* the frame read past its end or named an impossible codeword (the host decoder's -3 / -4) */
#define NYQ_ENT_TOO_LARGE 32
typedef struct nyq_ent_info {
    unsigned int range_final;        /* the range decoder's final state (OPUS_GET_FINAL_RANGE's CELT part) */
    short pf_pitch;                  /* post-filter period, tapset and gain index (gain = 0.09375 * index; 0: none) */
    unsigned char pf_tapset, pf_gain_index;
    unsigned char flags;             /* NYQ_ENT_* */
    unsigned char lm, channels, start, end, pad[3];
} nyq_ent_info;                      /* 16 bytes */
typedef struct nyq_ent_state {
    float energy[42], log_energy[42], log_energy2[42];   /* oldBandE, oldLogE, oldLogE2 (celt_decoder_clean.c:685-718) */
    unsigned int range;
    unsigned int valid;                                   /* set by the call; anything else on entry (e.g. zeros): the stream starts here */
    unsigned int errors;                                  /* frames in error so far */
} nyq_ent_state;                                          /* 516 bytes */
#define NYQ_ENT_ENERGY_BYTES 672
size_t nyq_celt_entropy_tables_bytes(void);
int nyq_celt_entropy_dev(nyq_ctx *ctx, int LM, const void *d_tables, const unsigned char *d_payload, size_t payload_bytes, const nyq_ent_desc *d_desc,
                         size_t nstreams, size_t nframes, int channels, void *d_sym, size_t slot_bytes, nyq_ent_info *d_info, void *d_energy,
                         nyq_ent_state *d_state, int fresh);
/* slot_bytes: bytes per record slot of d_sym; 0 = nyq_celt_symbol_bytes_lm(channels, LM) (what the host-record entry points take:
 * busy frames come back NYQ_ENT_TOO_LARGE); nyq_celt_entropy_slot_bytes(channels, LM) holds ANY frame (the records never leave
 * the device, so the larger slot costs HBM only) -- read such records with nyq_celt_shape_slots_dev. */
size_t nyq_celt_entropy_slot_bytes(int channels, int LM);
int nyq_celt_shape_slots_dev(nyq_ctx *ctx, int LM, const void *d_sym, size_t slot_bytes, float *d_freq, size_t nstreams, size_t nframes,
                             int channels);
/* d_info[n] -> the per-frame arrays nyq_celt_synth_dev / nyq_celt_post_dev / nyq_celt_chain_dev take (transient flags, post-filter
 * period, gain, tapset), on the device */
int nyq_celt_entropy_split_dev(nyq_ctx *ctx, const nyq_ent_info *d_info, size_t n, unsigned char *d_transient, int *d_pf_pitch,
                               float *d_pf_gain, int *d_pf_tapset);

/* The host-buffer form: nyq_celt_symbols_to_pcm_mapped with the frames' BYTES in place of records -- `bytes`
 * [nstreams][frames_per_stream][nyq_celt_byte_slot()] (a frame's bytes at the start of its slot), `frame_words`
 * [nstreams][frames_per_stream] = len | coded channels << 16 | end band << 24.  The entropy stage, the band shapes, synthesis
 * and post-filter all run on the device; per-frame arrays are not passed (they come out of the entropy stage).  `state`
 * (nyq_celt_state_floats, zero = streams that start here) carries the entropy stage's state too: its last
 * nstreams * sizeof(nyq_ent_state) / 4 floats are a nyq_ent_state per stream (`errors` = frames in error so far).
 * nyq_ctx_set_entropy_tables(ctx, block of nyqh_entropy_tables, its size) once per context before the first call. */
int nyq_ctx_set_entropy_tables(nyq_ctx *ctx, const void *host_tables, size_t bytes);
size_t nyq_celt_byte_slot(void);
int nyq_celt_bytes_to_pcm_mapped(nyq_ctx *ctx, int LM, const unsigned char *bytes, const unsigned *frame_words, float *out,
                                 const nyq_out_desc *desc, float *state, size_t nstreams, size_t nframes, int channels,
                                 size_t frames_per_stream);

/* libvorbis' mdct_backward (third_party/libvorbis/src/mdct.c:397-491) on `batch` rows: n/2 coefficients
 * in, n samples out per row, n a power of two in 64..8192 (every Vorbis block size).
 * d_in [batch][n/2], d_out [batch][n].  out[i] = sum_k in[k] cos(2 pi/n (i + 1/2 + n/4)(k + 1/2)):
 * no window, no overlap-add (libvorbis does those in block.c). */
int nyq_vorbis_imdct_batch_dev(nyq_ctx *ctx, int n, const float *d_in, float *d_out, size_t batch);

/* A measurement utility, not a decoder operator: plain device copies of `bytes` (a multiple of 16) from d_src to d_dst in
 * `form` 0 .. nyq_device_copy_forms() - 1 (grid-stride and chunk-per-wave float4 copies at several occupancies).  The row
 * kernels read and write equal byte counts, so the best of these on the box at hand is their practical ceiling; bench.py
 * reports it as roofline.measured_device_copy_GBps.  Asynchronous on the context stream. */
int nyq_device_copy_forms(void);
const char *nyq_device_copy_form_name(int form);
int nyq_device_copy_dev(nyq_ctx *ctx, void *d_dst, const void *d_src, size_t bytes, int form);

/* ---- host-buffer variants (synchronous: H2D, kernel, D2H through context scratch) ----
 * This is the shape of the reference's own FFI (host pointers in, host pointers out:
 * third_party/opus/celt/mdct.c:52-55).  nyq_imdct_batch / nyq_imdct_chain cut a large batch into pieces
 * and run the upload of piece k+1, the kernel of piece k and the download of piece k-1 concurrently on
 * three HIP streams.  Buffers from nyq_host_alloc (pinned) make those copies true DMA transfers;
 * pageable buffers are accepted and staged by the runtime. */
void *nyq_host_alloc(size_t bytes);   /* NULL on failure */
void nyq_host_free(void *p);
int nyq_ifft_batch(nyq_ctx *ctx, int nfft, const float *in, float *out, size_t batch);
int nyq_imdct_batch(nyq_ctx *ctx, int shift, const float *in, const float *carry,
                    float *fin, float *tail, size_t batch);
int nyq_imdct_chain(nyq_ctx *ctx, int shift, const float *in, const float *carry0,
                    float *pcm, float *tail_out, size_t nchains, size_t len);
int nyq_vorbis_imdct_batch(nyq_ctx *ctx, int n, const float *in, float *out, size_t batch);
int nyq_celt_synth(nyq_ctx *ctx, int LM, const float *freq, const unsigned char *transient,
                   float *pcm, float *state, size_t nstreams, size_t nframes, int channels);
/* freq[] -> interleaved PCM in one call: nyq_celt_synth_dev followed by nyq_celt_post_dev:
 * everything celt_decode_with_ec does after denormalise_bands (celt_decoder_clean.c:620-723).
 * Host buffers; layouts as in the two _dev functions; out [nstreams][nframes*N][channels].
 * `state` (host, in/out, nyq_celt_state_floats() floats) carries the decoders from one call to the
 * next, e.g. across a change of frame size: [nsc][60] overlap carry, [nsc][1088] filtered history,
 * [nsc] de-emphasis memory, [nstreams][6] post-filter state (nsc = nstreams*channels), then [nstreams] nyq_ent_state (the
 * entropy stage's energies, range and error count: read and written by nyq_celt_bytes_to_pcm_mapped only; the other entry
 * points leave it alone).  NULL means freshly reset decoders and discards the final state. */
size_t nyq_celt_state_floats(size_t nstreams, int channels);
int nyq_celt_frames_to_pcm(nyq_ctx *ctx, int LM, const float *freq, const unsigned char *transient,
                           const int *pf_pitch, const float *pf_gain, const int *pf_tapset,
                           float *out, float *state, size_t nstreams, size_t nframes, int channels);

/* The same on a WINDOW of longer per-stream host arrays: consecutive streams are `frames_per_stream` (>= nframes)
 * frames apart in freq / transient / pf_* / out, and the pointers address the window's first frame of the first
 * stream.  With `state` carried from call to call this decodes long streams slice by slice, many streams at a
 * time, in bounded device memory -- bit-identical to one call over the whole length. */
int nyq_celt_frames_to_pcm_window(nyq_ctx *ctx, int LM, const float *freq, const unsigned char *transient,
                                  const int *pf_pitch, const float *pf_gain, const int *pf_tapset,
                                  float *out, float *state, size_t nstreams, size_t nframes, int channels,
                                  size_t frames_per_stream);

/* nyq_celt_frames_to_pcm_window with per-stream destinations: desc [nstreams] (HOST array; its `base` pointers are
 * device memory) or NULL.  Streams with a destination are written there by the kernels and are not downloaded into
 * `out`; when every stream has one, `out` may be NULL. */
int nyq_celt_frames_to_pcm_mapped(nyq_ctx *ctx, int LM, const float *freq, const unsigned char *transient,
                                  const int *pf_pitch, const float *pf_gain, const int *pf_tapset,
                                  float *out, const nyq_out_desc *desc, float *state, size_t nstreams, size_t nframes,
                                  int channels, size_t frames_per_stream);

/* The same with SYMBOL records in place of freq[] (`sym` [nstreams][frames_per_stream][nyq_celt_symbol_bytes_lm(channels, LM)]): the
 * band shapes are built on the device (nyq_celt_shape_dev), then the chain runs as above.  desc may be NULL (dense `out`). */
int nyq_celt_symbols_to_pcm_mapped(nyq_ctx *ctx, int LM, const void *sym, const unsigned char *transient,
                                   const int *pf_pitch, const float *pf_gain, const int *pf_tapset,
                                   float *out, const nyq_out_desc *desc, float *state, size_t nstreams, size_t nframes,
                                   int channels, size_t frames_per_stream);

/* The same with the records PACKED back to back (a record is as long as its content: head, gains, nops operations, nvecs
 * vectors, nleaves leaves [, levels] rounded up to 16 bytes -- about half a slot on music): stream s's records start at
 * (char *)sym + s * stream_bytes, frame f of the call at 16 * offsets[s * (frames_per_stream + 1) + f] bytes from there and ends where
 * frame f + 1 begins (so `offsets` has one entry more per stream than frames; for a time slice pass the pointer to the slice's
 * first entry, `sym` unchanged).  offsets == NULL: slots, as above. */
int nyq_celt_symbols_packed_to_pcm_mapped(nyq_ctx *ctx, int LM, const void *sym, const unsigned int *offsets, size_t stream_bytes,
                                          const unsigned char *transient, const int *pf_pitch, const float *pf_gain,
                                          const int *pf_tapset, float *out, const nyq_out_desc *desc, float *state,
                                          size_t nstreams, size_t nframes, int channels, size_t frames_per_stream);

/* ---- the reference's operator boundary, kept verbatim ------------------- */
/* cuda/mdct_cuda.hpp:89-91 (impl mdct_cuda.cu:314-392).  Host pointers, caller-owned.
 * N = mdct.n >> shift (already shifted, mdct.c:249-253); input element k at
 * input[k*stride]; output is read-modify-write over N/2 + overlap/2 floats with the
 * carry in output[0..overlap/2); trig/window are the mode's tables. */
void processMDCTCuda(const float *input, float *output, const float *trig, int N, int shift,
                     int stride, float sine, int overlap, const float *window);
/* cuda/mdct_cuda.hpp:92-94 (impl mdct_cuda.cu:562-584): two channels per call. */
void processMDCTCudaB1C2(const float *input[2], float *output[2], const float *trig, int N,
                         int shift, int stride, float sine, int overlap, const float *window);
/* cuda/mdct_cuda.hpp:96-98: declared, never called, and defined with another signature in the reference
 * (mdct_cuda_b8.cu:482-501).  Exported so that the header's whole symbol set links; it does what the DECLARATION says:
 * eight rows of one size per call (= four B1C2 calls), one launch. */
void processMDCTCudaB8C2(const float *input[8], float *output[8], const float *trig, int N,
                         int shift, int stride, float sine, int overlap, const float *window);
/* cuda/mdct_cuda.hpp:100, called from examples/src/Main.cpp:127-129 */
void cleanupCudaBuffers(void);
/* Not in the reference.  The void operators above cannot return a status: by default a failure (no device, a HIP error,
 * a call outside the static 48 kHz mode) prints and ends the process, as the reference's offload does (mdct_cuda.cu:11-19).
 * With a handler installed the handler is called instead (entry point, reason) and, when it returns, the call returns with
 * `output` untouched -- the integrator decides what to do with the stream.  The handler runs with no library lock held: it
 * may call any entry point (the operators and cleanupCudaBuffers included).  After a HIP failure the operators' context is
 * dropped and the next call creates a new one.  NULL restores the default.  No CPU fallback. */
void nyq_shim_set_error_handler(void (*handler)(const char *who, const char *what));
/* cuda/mdct_cuda.hpp:83, called from examples/src/Main.cpp:26-28 */
void printCudaVersion(void);

#ifdef __cplusplus
}
#endif
#endif /* NYQ_IMDCT_H */
