// nyq_shape_kernel.hpp -- the band shapes of CELT frames built ON THE DEVICE from the symbols the host's entropy stage read.
//
// quant_all_bands' arithmetic (bands.c:1355-1518) -- pulse vectors to unit-norm coefficients (vq.c normalise_residual), the
// spreading rotation (vq.c:40-111), folding and noise filling, the Haar / Hadamard resolution changes (bands.c haar1,
// (de)interleave_hadamard), mid / side merging (bands.c:391-441) -- and denormalise_bands (bands.c:192-256) need no bit of the
// stream once the symbols are known: the host (libnyquist_amd/host/celt_decoder.cpp, phase 1) sends, per frame, the leaves of
// every band's split tree with the CODEWORD of each pulse vector, one record per band vector and a short list of operations in
// execution order (include/nyq_imdct.h: nyq_sym_*), and this kernel (a) unranks every pulse vector of the frame (cwrs.c
// decode_pulses), a leaf per lane, (b) executes the list -- collapse masks, fill decisions and all the float work; what the
// host's own phases 1b and 2 do (same order of the dependent operations: the rotation recurrences, the noise generator, the fold
// sources; reductions are summed in another order).  Half of the host's per-frame time moves here, where it is one wavefront
// per frame among thousands.
//
// One wavefront = one frame at a time (frames handed out by a counter; kShapeWaves frames side by side in a workgroup, sharing
// the codebook table).  The frame's coefficients X[C][960] and the fold memory live in the wave's LDS slice; its record is read
// in place.  Pass A, a leaf per LANE: everything a pulse leaf needs is its own -- codeword -> pulse counts ->
// unit-norm coefficients -> spreading rotation, all in place in X, the chains' carried values in registers and the loads
// of a chain issued four steps ahead (a step is otherwise one LDS round trip).  Pass B, wave-uniform: the operation list in
// order -- collapse masks, fills from the bands below, resolution changes, stereo -- lanes over the bins.
#pragma once
#include <hip/hip_runtime.h>

#include <cstdint>

#include "nyq_kernels.hpp"

namespace nyq {

// (layout of include/nyq_imdct.h: nyq_sym_head / leaf / vec / op)
struct SymHead {
    unsigned seed;
    unsigned short nleaves, nvecs, nops;
    unsigned char flags, spread, start, end, channels, lm;
    unsigned reserved[4];
};
struct SymLeaf {
    short off, n, k;
    unsigned char blocks, kind;
    float gain;
    short fold_off;
    unsigned char shift, pad;
    unsigned index;
    short abs, pad2;
    unsigned short img[8];
};
struct SymVec {
    short x, n, fold, out, nb_tree, leaf0, leaf1;
    unsigned char sel, recombine, time_divide, b_tree, b_in, band, cm_ch, fill_mode, fill_lo, fill_hi;
};
struct SymOp {
    unsigned char kind, band;
    short a, b, n;
    float f0, f1;
};
constexpr int kSymMaxOps = 113, kSymMaxVecs = 44;
constexpr int kSymN = 960;                                           // bins per channel of a 20 ms frame (LM 3); 120 << LM in general
// a record is COMPACT: head | log_gain[42] | ops[nops] | vecs[nvecs] | leaves[nleaves] | (anti-collapse) level[42], or head | freq[];
// the slot of the fixed-stride form (sym_bytes) holds the largest record the entropy stage writes
constexpr int kSymOpsOff = 32 + 42 * 4;                              // 200
constexpr int kSymFreqOff = 32;
constexpr int kSymSlotFixed = 3072;
// leaves of one frame: a vector of n bins splits into at most min(2^(LM+1), n/2) leaves -- 208 per channel at LM 3 (8 bands x 4,
// 4 x 8, 9 x 16), fewer below; the host's records stop at 192 (what their slot holds), the device's own entropy stage at this
constexpr int kSymMaxLeaves = 416;
constexpr int kPvqDim = 178;                                         // U(n, k) for n, k < 178 (the widest band has 176 bins)
// slot of a frame of 120 << LM samples: the 20 ms slot, 5/8 of it at 10 ms, and a floor below that (the fixed parts, a
// leaf or two per vector: short frames spend their bits on fewer, not smaller, records); never less than head + freq[]
__host__ __device__ inline size_t sym_bytes(int channels, int LM = 3) {
    const size_t full = (size_t)kSymSlotFixed + (size_t)channels * kSymN * 4;
    const size_t floor_ = LM == 3 ? 0 : (size_t)2048 * (size_t)channels + 512;
    const size_t scaled = LM == 3 ? full : LM == 2 ? full * 5 / 8 : full >> (3 - LM);
    return ((scaled > floor_ ? scaled : floor_) + 15) & ~(size_t)15;
}

constexpr int kShapeNorm = 2 * 800;                                  // fold memory: two channels x bins below the last band
constexpr int kShapeLdsFloats = 2 * kSymN + kShapeNorm + 192 + 192;

// U(n, k) of cwrs.c (cwrs.c:206-440: the number of k-pulse vectors in n dimensions whose first coordinate is not negative
// ...), as 32-bit words.  A valid codeword only ever meets entries below 2^32 -- 2595 of the 178 x 178 -- so the table is kept
// COMPACT: row k holds U(0 .. len_k - 1, k), everything beyond reads as 2^32 - 1 ("larger than any codeword"); 10.4 KB, which
// a workgroup keeps in LDS (the unranking walk is a chain of dependent look-ups: LDS latency, not L2 latency, per step).
// Layout: info[kPvqDim] = offset | len << 16, then the rows.  Built once per context on the host (shape_core).
constexpr int kPvqWords = 2595, kPvqInfo = 180;
inline int pvq_table_build(unsigned *out /* kPvqInfo + kPvqWords */) {
    static unsigned long long U[kPvqDim][kPvqDim];
    const unsigned long long cap = ~0ull >> 1;
    for (int n = 0; n < kPvqDim; n++)
        for (int k = 0; k < kPvqDim; k++) {
            if (n == 0 || k == 0) {
                U[n][k] = (n == 0 && k == 0) ? 1 : 0;
                continue;
            }
            unsigned long long v = U[n - 1][k] + U[n][k - 1];
            if (v > cap) v = cap;
            v += U[n - 1][k - 1];
            if (v > cap) v = cap;
            U[n][k] = v;
        }
    int off = 0;
    for (int k = 0; k < kPvqInfo; k++) {
        int len = 0;
        while (k < kPvqDim && len < kPvqDim && U[len][k] < 0x100000000ull) len++;   // (U grows with n: a prefix)
        out[k] = (unsigned)off | (unsigned)len << 16;
        for (int n = 0; n < len; n++) {
            if (off < kPvqWords) out[kPvqInfo + off] = (unsigned)U[n][k];
            off++;
        }
    }
    return off;                                                      // == kPvqWords
}

// codeword -> pulse vector y[0 .. n) (cwrs.c cwrsi): one lane, its leaf.  y: the leaf's slots in X (the pulse counts are
// left there as integers; the caller turns them into coefficients in place).  Returns |y|^2; *cm = which of the leaf's
// `blocks` interleaved short blocks received a pulse.  While no pulse is found the row pair (k, k + 1) stands still and the
// look-ups of the next coordinates do not depend on the codeword: they are issued four coordinates ahead.
__device__ __forceinline__ unsigned pvq_at(const unsigned *tab, unsigned info, int n) {
    return n < (int)(info >> 16) ? tab[kPvqInfo + (info & 0xffffu) + n] : 0xFFFFFFFFu;
}
__device__ __forceinline__ int pvq_unrank(const unsigned *tab, int n, int k, unsigned idx, int *y, int blocks, unsigned *cm) {
    const int per = blocks > 1 ? n / blocks : n;
    int yy = 0, j = 0;
    unsigned mask = 0;
    while (n > 2) {
        const unsigned info0 = tab[k], info1 = tab[k + 1];
        unsigned hi[4], lo[4];
#pragma unroll
        for (int q = 0; q < 4; q++) {
            hi[q] = pvq_at(tab, info1, n - q);                       // U(n - q, k + 1)
            lo[q] = pvq_at(tab, info0, n - q);                       // U(n - q, k)
        }
#pragma unroll
        for (int q = 0; q < 4; q++) {
            if (n <= 2) break;
            const bool neg = idx >= hi[q];
            if (neg) idx -= hi[q];
            int v = 0;
            bool moved = false;
            if (lo[q] <= idx) {
                idx -= lo[q];
            } else {
                int kk = k;
                if (kk > n && pvq_at(tab, tab[n], n) > idx) kk = n;
                unsigned p;
                do {
                    kk--;
                    p = pvq_at(tab, tab[kk], n);
                } while (p > idx);
                idx -= p;
                v = k - kk;
                k = kk;
                moved = true;
            }
            y[j] = neg ? -v : v;
            yy += v * v;
            if (v) mask |= 1u << (j / per);
            j++;
            n--;
            if (moved) break;                                        // (another row pair: look up again)
        }
    }
    {
        const unsigned a = 2 * (unsigned)k + 1;
        const bool neg = idx >= a;
        if (neg) idx -= a;
        const int kk = (int)((idx + 1) >> 1);
        if (kk) idx -= 2 * (unsigned)kk - 1;
        const int v = k - kk;
        y[j] = neg ? -v : v;
        y[j + 1] = idx ? -kk : kk;
        yy += v * v + kk * kk;
        if (v) mask |= 1u << (j / per);
        if (kk) mask |= 1u << ((j + 1) / per);
    }
    *cm = blocks > 1 ? mask : 1u;
    return yy;
}

__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
    for (int m = 32; m >= 1; m >>= 1) v += __shfl_xor(v, m);
    return v;
}
// the noise generator m steps ahead: x -> A^m x + C (A^(m-1) + ... + 1)  (bands.c:61-64 iterated)
__device__ __forceinline__ unsigned lcg_jump(unsigned seed, int m) {
    unsigned a = 1664525u, c = 1013904223u, ra = 1u, rc = 0u;        // (ra, rc): the map applied so far
    while (m > 0) {
        if (m & 1) {
            ra = ra * a;
            rc = rc * a + c;
        }
        c = c * a + c;
        a = a * a;
        m >>= 1;
    }
    return ra * seed + rc;
}

__device__ __forceinline__ void shape_haar(float *x, int n0, int stride, int lane) {
    const int half = n0 >> 1, total = half * stride;
    for (int t = lane; t < total; t += kWave) {
        const int j = t / stride, i = t - j * stride;
        float *p = x + stride * 2 * j + i;
        const float a = .70710678f * p[0], b = .70710678f * p[stride];
        p[0] = a + b;
        p[stride] = a - b;
    }
    NYQ_WAVE_SYNC();
}

__device__ __forceinline__ void shape_regroup(float *x, float *tmp, int n0, int stride, bool hadamard, bool toBlocks, int lane) {
    const int order[30] = {1, 0, 3, 0, 2, 1, 7, 0, 4, 3, 6, 1, 5, 2, 15, 0, 8, 7, 12, 3, 11, 4, 14, 1, 9, 6, 13, 2, 10, 5};
    const int n = n0 * stride;
    for (int t = lane; t < n; t += kWave) {
        const int i = t / n0, j = t - i * n0;
        const int blk = hadamard ? order[stride - 2 + i] : i;
        if (toBlocks) tmp[blk * n0 + j] = x[j * stride + i];
        else tmp[j * stride + i] = x[blk * n0 + j];
    }
    NYQ_WAVE_SYNC();
    for (int t = lane; t < n; t += kWave) x[t] = tmp[t];
    NYQ_WAVE_SYNC();
}

// One rotation pass (vq.c exp_rotation1) over one block of `len` floats, by ONE lane: the residues modulo `stride` are
// independent chains, each walked forward then backward with the carried value in a register; the other operand of a step
// is untouched data, so its loads are issued four steps ahead of the arithmetic.
__device__ __forceinline__ void lane_rotate(float *p, int len, int stride, float c, float s) {
#pragma clang fp contract(off)
    const int top = len - 2 * stride - 1;
    for (int r = 0; r < stride; r++) {
        if (r < len - stride) {                                      // forward: i = r, r + stride, ... < len - stride
            float a = p[r];
            int i = r;
            for (; i + 3 * stride < len - stride; i += 4 * stride) {
                const float x0 = p[i + stride], x1 = p[i + 2 * stride], x2 = p[i + 3 * stride], x3 = p[i + 4 * stride];
                float h;
                h = c * x0 + s * a; p[i] = c * a - s * x0; a = h;
                h = c * x1 + s * a; p[i + stride] = c * a - s * x1; a = h;
                h = c * x2 + s * a; p[i + 2 * stride] = c * a - s * x2; a = h;
                h = c * x3 + s * a; p[i + 3 * stride] = c * a - s * x3; a = h;
            }
            for (; i < len - stride; i += stride) {
                const float x2 = p[i + stride];
                const float h = c * x2 + s * a;
                p[i] = c * a - s * x2;
                a = h;
            }
            p[i] = a;
        }
        if (r <= top) {                                              // backward: from the largest i = r (mod stride) <= top
            int i = top - ((top - r) % stride);
            float b = p[i + stride];
            for (; i - 3 * stride >= 0; i -= 4 * stride) {
                const float x0 = p[i], x1 = p[i - stride], x2 = p[i - 2 * stride], x3 = p[i - 3 * stride];
                p[i + stride] = c * b + s * x0; b = c * x0 - s * b;
                p[i] = c * b + s * x1; b = c * x1 - s * b;
                p[i - stride] = c * b + s * x2; b = c * x2 - s * b;
                p[i - 2 * stride] = c * b + s * x3; b = c * x3 - s * b;
            }
            for (; i >= 0; i -= stride) {
                const float x1 = p[i];
                p[i + stride] = c * b + s * x1;
                b = c * x1 - s * b;
            }
            p[i + stride] = b;
        }
    }
}

// a pulse leaf, start to finish, by one lane: codeword -> pulse counts -> gain / |y| -> the spreading rotation (vq.c:65-111,
// decoder direction: the long-stride pass with (s, c) first, then stride 1 with (c, s)), every interleaved block a chain
__device__ __forceinline__ void lane_pulse_leaf(const unsigned *tab, const SymLeaf &lf, float *X, int spread, unsigned *cm) {
#pragma clang fp contract(off)
    float *o = X + lf.abs;
    int *y = reinterpret_cast<int *>(o);
    const int n = lf.n;
    const int yy = pvq_unrank(tab, n, lf.k, lf.index, y, lf.blocks, cm);
    const float g = (1.f / sqrtf((float)yy)) * lf.gain;
    for (int j = 0; j < n; j++) o[j] = g * (float)y[j];
    if (spread == 0 || 2 * lf.k >= n) return;
    const int factor = spread == 1 ? 15 : spread == 2 ? 10 : 5;
    const float gn = (float)(1.0f * n) / (float)(n + factor * lf.k);
    const float theta = .5f * (gn * gn);
    const float c = cosf((.5f * 3.141592653f) * theta);              // (the host rounds a double cosine: the same to an ulp)
    const float s = cosf((.5f * 3.141592653f) * (1.0f - theta));
    const int stride = lf.blocks, len = n / stride;
    int stride2 = 0;
    if (n >= 8 * stride) {
        stride2 = 1;
        while ((stride2 * stride2 + stride2) * stride + (stride >> 2) < n) stride2++;
    }
    for (int b = 0; b < stride; b++) {
        if (stride2) lane_rotate(o + b * len, len, stride2, s, c);
        lane_rotate(o + b * len, len, 1, c, s);
    }
}

struct ShapeFrame {                  // (all in the wave's LDS slice)
    const SymOp *ops;
    const SymVec *vecs;
    const SymLeaf *leaves;
    const unsigned short *leafCm;    // collapse mask of a pulse leaf
    unsigned char *masks;            // [2][21] collapse masks of the bands so far
};

__device__ __forceinline__ void shape_vector(const ShapeFrame &F, const SymVec v, float *X, float *norm, float *norm2, float *work,
                                             float *tmp, unsigned &seed, int lane) {
#pragma clang fp contract(off)
    float *x = X + v.x;
    const int n = v.n, recombine = v.recombine, timeDivide = v.time_divide, Btree = v.b_tree;
    const bool longBlocks = v.b_in == 1;
    const float *src = v.fold >= 0 ? (v.sel ? norm2 : norm) + v.fold : nullptr;
    // the vector's initial fill mask: the collapse masks of the bands it may fold from (bands.c:1455-1481)
    unsigned fill0 = 0;
    if (v.fill_mode == 3) {
        fill0 = (1u << v.b_in) - 1;
    } else {
        for (int f = v.fill_lo; f < v.fill_hi; f++)
            fill0 |= v.fill_mode == 1 ? F.masks[f] : v.fill_mode == 2 ? F.masks[21 + f] : (unsigned)(F.masks[f] | F.masks[21 + f]);
    }
    // what becomes of the leaves without pulses, and the vector's collapse mask (every lane computes the same)
    unsigned cm = 0, kinds = 0;                                      // kinds: 2 bits per leaf of the vector (at most 16)
    bool folds = false;
    for (int l = v.leaf0; l < v.leaf1; l++) {
        const SymLeaf &lf = F.leaves[l];
        unsigned lcm, kind = 0;
        if (lf.kind == 0) {
            lcm = F.leafCm[l];
        } else {
            unsigned fill = 0;
#pragma unroll
            for (int i = 0; i < 8; i++)
                if (fill0 >> i & 1) fill |= lf.img[i];
            if (!fill) {
                kind = 1;
                lcm = 0;
            } else if (lf.fold_off < 0) {
                kind = 2;
                lcm = (1u << lf.blocks) - 1;
            } else {
                kind = 3;
                lcm = fill;
                folds = true;
            }
        }
        kinds |= kind << (2 * (l - v.leaf0));
        cm |= lcm << lf.shift;
    }
    {
        int Bm = Btree;
        for (int k = 0; k < timeDivide; k++) {
            Bm >>= 1;
            cm |= cm >> Bm;
        }
        for (int k = 0; k < recombine; k++) {                       // (bit_deinterleave_table: every bit doubled)
            const unsigned c4 = cm & 0xF;
            cm = (c4 & 1) * 3 | (c4 >> 1 & 1) * 0xC | (c4 >> 2 & 1) * 0x30 | (c4 >> 3 & 1) * 0xC0;
        }
        Bm <<= recombine;
        cm &= (1u << Bm) - 1;
        NYQ_WAVE_SYNC();
        if (lane == 0) {
            if (v.cm_ch & 1) F.masks[v.band] |= (unsigned char)cm;
            if (v.cm_ch & 2) F.masks[21 + v.band] |= (unsigned char)cm;
        }
        NYQ_WAVE_SYNC();
    }
    if (folds && (recombine || timeDivide || Btree > 1)) {
        for (int j = lane; j < n; j += kWave) work[j] = src[j];
        NYQ_WAVE_SYNC();
        for (int k = 0; k < recombine; k++) shape_haar(work, n >> k, 1 << k, lane);
        int bb = v.b_in >> recombine, nn = (n / v.b_in) << recombine;
        for (int k = 0; k < timeDivide; k++) {
            shape_haar(work, nn, bb, lane);
            bb <<= 1;
            nn >>= 1;
        }
        if (Btree > 1) shape_regroup(work, tmp, v.nb_tree >> recombine, Btree << recombine, longBlocks, true, lane);
        src = work;
    }
    // the leaves without pulses, in tree order (the noise generator advances through the filled ones in this order); the
    // pulse leaves were built by pass A
    for (int l = v.leaf0; l < v.leaf1; l++) {
        const unsigned kind = kinds >> (2 * (l - v.leaf0)) & 3;
        if (kind == 0) continue;
        const SymLeaf &lf = F.leaves[l];
        float *o = x + lf.off;
        const int ln = lf.n;
        if (kind == 1) {
            for (int j = lane; j < ln; j += kWave) o[j] = 0.f;
            NYQ_WAVE_SYNC();
        } else {
            float e = 0.f;
            for (int j = lane; j < ln; j += kWave) {
                const unsigned sj = lcg_jump(seed, j + 1);
                const float val = kind == 2 ? (float)((int)sj >> 20) : src[lf.fold_off + j] + ((sj & 0x8000u) ? 1.0f / 256 : -1.0f / 256);
                o[j] = val;
                e += val * val;
            }
            seed = lcg_jump(seed, ln);
            const float E = wave_sum(e) + 1e-15f;
            const float g = (1.f / sqrtf(E)) * lf.gain;
            NYQ_WAVE_SYNC();
            for (int j = lane; j < ln; j += kWave) o[j] = g * o[j];
            NYQ_WAVE_SYNC();
        }
    }
    if (Btree > 1) shape_regroup(x, tmp, v.nb_tree >> recombine, Btree << recombine, longBlocks, false, lane);
    int B = Btree, nb = v.nb_tree;
    for (int k = 0; k < timeDivide; k++) {
        B >>= 1;
        nb <<= 1;
        shape_haar(x, nb, B, lane);
    }
    for (int k = 0; k < recombine; k++) shape_haar(x, n >> k, 1 << k, lane);
    if (v.out >= 0) {
        float *dst = (v.sel ? norm2 : norm) + v.out;
        const float g = sqrtf((float)n);
        for (int j = lane; j < n; j += kWave) dst[j] = g * x[j];
        NYQ_WAVE_SYNC();
    }
}

// band edges of the 48 kHz mode (modes.c:41-44) at 120 << LM samples per frame
__device__ __forceinline__ int shape_edge(int i, int LM) {
    const short e[22] = {0, 1, 2, 3, 4, 5, 6, 7, 8, 10, 12, 14, 16, 20, 24, 28, 34, 40, 48, 60, 78, 100};
    return e[i] << LM;
}

// measurement switches (tools/shape_time.py builds variants with them; the product defines none)
// NYQ_SHAPE_STAGE 0 (the product): the frame's operations / vectors / leaves are read where they lie, in global memory -- pass B's
// addresses are wave-uniform, i.e. scalar loads -- and NINE waves per CU fit beside the table; 1: they are staged into LDS first
// (10.5 KB per wave: five waves per CU).  Measured on 89,472 real frames: 5.75 ms against 8.86 (profiles/r04_av_*): a frame
// waits longer for its records, and nearly twice as many frames wait at once.
#ifndef NYQ_SHAPE_STAGE
#define NYQ_SHAPE_STAGE 0
#endif
#ifndef NYQ_SHAPE_WAVES
#define NYQ_SHAPE_WAVES (NYQ_SHAPE_STAGE ? 5 : 9)
#endif
#ifndef NYQ_SHAPE_DBG_NO_A
#define NYQ_SHAPE_DBG_NO_A 0
#endif
#ifndef NYQ_SHAPE_DBG_NO_B
#define NYQ_SHAPE_DBG_NO_B 0
#endif
constexpr int kShapeWaves = NYQ_SHAPE_WAVES;                         // frames side by side in a workgroup (they share the table)
struct ShapeWaveLds {
    float f[kShapeLdsFloats];                                        // X | fold memory | two work vectors
    // the frame's operations, vectors and leaves, staged as they lie in the record (one contiguous range)
#if NYQ_SHAPE_STAGE
    unsigned rec[(kSymMaxOps * sizeof(SymOp) + kSymMaxVecs * sizeof(SymVec) + kSymMaxLeaves * sizeof(SymLeaf)) / 4];
#endif
    unsigned short leafCm[kSymMaxLeaves];
    unsigned char masks[2 * 21 + 6];
};
static_assert(sizeof(SymLeaf) % 4 == 0 && sizeof(SymOp) % 4 == 0 && sizeof(SymVec) % 4 == 0, "staged word by word");

__device__ __forceinline__ void stage_words(void *dst, const void *src, int bytes, int lane) {
    unsigned *d = static_cast<unsigned *>(dst);
    const unsigned *s = static_cast<const unsigned *>(src);
    for (int w = lane; w < bytes / 4; w += kWave) d[w] = s[w];
}

template <int LM>   // frames of 120 << LM samples (an instance per size: band edges, block counts and strides are constants)
__global__ __launch_bounds__(kWave *kShapeWaves) void celt_shape_kernel(const unsigned char *__restrict__ sym, float *__restrict__ freq,
                                                                        const unsigned *__restrict__ pvq, long nstreams, long nframes,
                                                                        int channels, long sstride, long fstride,
                                                                        const unsigned *__restrict__ offsets, long ostride,
                                                                        unsigned *__restrict__ next_frame, long slot) {
#pragma clang fp contract(off)
    __shared__ __attribute__((aligned(16))) ShapeWaveLds wl[kShapeWaves];
    __shared__ unsigned tab[kPvqInfo + kPvqWords];
    for (int i = threadIdx.x; i < kPvqInfo + kPvqWords; i += kWave * kShapeWaves) tab[i] = pvq[i];
    __syncthreads();
    const int lane = threadIdx.x & (kWave - 1), wv = threadIdx.x >> 6;
    ShapeWaveLds &L = wl[wv];
    float *X = L.f, *norm = L.f + 2 * kSymN, *work = norm + kShapeNorm, *tmp = work + 192;
    const size_t rec = (size_t)slot;                                 // bytes per record slot (sym_bytes(channels, LM) unless the caller says otherwise)
    constexpr int N = 120 << LM;                                     // bins per channel
    const long total = nstreams * nframes;
    // Frames are handed out by a counter (zeroed before the launch), not dealt in advance: a transient frame costs several
    // times a plain one, and a wave that drew three of them would be what the launch waits for.  Every wave leaves the
    // loop at the first index beyond the last frame.
    for (;;) {
        unsigned fetched = 0;
        if (lane == 0) fetched = atomicAdd(next_frame, 1u);
        const long u = (long)__builtin_amdgcn_readfirstlane(fetched);
        if (u >= total) break;
        const long s = u / nframes, f = u - s * nframes;
        // where the frame's record is: packed back to back inside its stream's region (offsets[stream * ostride + frame], 16-byte
        // units from the region's start) or in slots of sym_bytes; a stream's region is sstride slots long either way
        const unsigned char *r = sym + (size_t)s * (size_t)sstride * rec + (offsets ? (size_t)offsets[s * ostride + f] * 16 : (size_t)f * rec);
        float *out = freq + ((size_t)s * (size_t)fstride + (size_t)f) * (size_t)channels * N;
        const SymHead H = *reinterpret_cast<const SymHead *>(r);
        const int CC = channels;                                     // the stream's channels: the layout of freq[]
        if (H.flags & 1) {                                           // the host built this frame itself: its freq[] is the body
            const float4 *b4 = reinterpret_cast<const float4 *>(r + kSymFreqOff);
            float4 *o4 = reinterpret_cast<float4 *>(out);
            for (int j = lane; j < CC * N / 4; j += kWave) o4[j] = b4[j];
            continue;
        }
        if (H.nops == 0) {                                           // a record of zeros: a silent (or padding) frame
            float4 *o4 = reinterpret_cast<float4 *>(out);
            for (int j = lane; j < CC * N / 4; j += kWave) o4[j] = float4{0.f, 0.f, 0.f, 0.f};
            continue;
        }
        const int C = H.channels == 2 ? 2 : 1;                       // what the packet codes
        if (H.lm != LM || H.start > 20 || H.end > 21 || H.start > H.end || H.nops > kSymMaxOps || H.nvecs > kSymMaxVecs || H.nleaves > kSymMaxLeaves) {
            // (a damaged head -- its counts would place the record's parts outside the record: a silent frame)
            float4 *o4 = reinterpret_cast<float4 *>(out);
            for (int j = lane; j < CC * N / 4; j += kWave) o4[j] = float4{0.f, 0.f, 0.f, 0.f};
            continue;
        }
        const int nleaves = H.nleaves < kSymMaxLeaves ? H.nleaves : kSymMaxLeaves;
        const int nops = H.nops < kSymMaxOps ? H.nops : kSymMaxOps, nvecs = H.nvecs < kSymMaxVecs ? H.nvecs : kSymMaxVecs;
        // band gains: 2^(energy + mean), celt_exp2 of the float build (a double exponential rounded to float: mathops.h), a band
        // per lane -- the host sends the exponents
        const float laneGain = lane < 2 * 21 ? (float)exp(0.6931471805599453094 * (double)reinterpret_cast<const float *>(r + 32)[lane]) : 0.f;
        NYQ_WAVE_SYNC();                                             // (the previous frame's last reads of this slice)
        // (the counts were checked against their bounds above: the three parts are one contiguous range of the record)
        const int bops = nops * (int)sizeof(SymOp), bvecs = nvecs * (int)sizeof(SymVec), bleaves = nleaves * (int)sizeof(SymLeaf);
        // a SPREAD record (the device's own entropy stage writes these: nyq_entropy_core.hpp) names where its lists are in the
        // head's reserved words; zero = the compact form
        int opsOff = kSymOpsOff, vecsOff = kSymOpsOff + bops, leavesOff = kSymOpsOff + bops + bvecs, levelOff = kSymOpsOff + bops + bvecs + bleaves;
        if (H.reserved[0] | H.reserved[1]) {
            opsOff = (int)(H.reserved[0] & 0xffffu);
            vecsOff = (int)(H.reserved[0] >> 16);
            leavesOff = (int)(H.reserved[1] & 0xffffu);
            levelOff = (int)(H.reserved[1] >> 16);
            const bool fits = !offsets && !NYQ_SHAPE_STAGE && ((opsOff | vecsOff | leavesOff | levelOff) & 7) == 0 && opsOff >= kSymOpsOff &&
                              (size_t)(opsOff + bops) <= rec && (size_t)(vecsOff + bvecs) <= rec && (size_t)(leavesOff + bleaves) <= rec &&
                              (size_t)(levelOff + 2 * 21 * 4) <= rec;
            if (!fits) {
                float4 *o4 = reinterpret_cast<float4 *>(out);
                for (int j = lane; j < CC * N / 4; j += kWave) o4[j] = float4{0.f, 0.f, 0.f, 0.f};
                continue;
            }
        }
        const unsigned char *rleaves = r + leavesOff;
#if NYQ_SHAPE_STAGE
        stage_words(L.rec, r + kSymOpsOff, bops + bvecs + bleaves, lane);
        const SymOp *Lops = reinterpret_cast<const SymOp *>(L.rec);
        const SymVec *Lvecs = reinterpret_cast<const SymVec *>(L.rec + bops / 4);
        const SymLeaf *Lleaves = reinterpret_cast<const SymLeaf *>(L.rec + (bops + bvecs) / 4);
#else
        const SymOp *Lops = reinterpret_cast<const SymOp *>(r + opsOff);
        const SymVec *Lvecs = reinterpret_cast<const SymVec *>(r + vecsOff);
        const SymLeaf *Lleaves = reinterpret_cast<const SymLeaf *>(rleaves);
#endif
        if (lane < 2 * 21) L.masks[lane] = 0;
        NYQ_WAVE_SYNC();
        ShapeFrame F;
        F.ops = Lops;
        F.vecs = Lvecs;
        F.leaves = Lleaves;
        F.leafCm = L.leafCm;
        F.masks = L.masks;
        float *norm2 = norm + (shape_edge(20, LM) - shape_edge(H.start, LM));
        unsigned seed = H.seed;
        // Validation, a record per lane.  The records come from this project's own entropy stage, but the ABI takes them from
        // any caller: every offset and count the passes below use is checked here against its bound, once and in parallel; a
        // frame with one record that cannot be is played as silence (no loop below is unbounded, no write leaves the wave's
        // own working set, whatever the bytes).
        {
            bool bad = false;
            for (int l = lane; l < nleaves; l += kWave) {
                const SymLeaf &lf = Lleaves[l];
                bad |= lf.n < 1 || lf.n > 176 || lf.off < 0 || lf.off + lf.n > 176 || lf.blocks < 1 || lf.blocks > 16 || lf.shift > 15 ||
                       lf.abs < 0 || lf.abs + lf.n > 2 * N || lf.fold_off + lf.n > 176 || lf.kind > 1 ||
                       (lf.kind == 0 && (lf.n < 2 || lf.k < 1 || lf.k > 176 || lf.n % lf.blocks != 0));
            }
            for (int q = lane; q < nvecs; q += kWave) {
                const SymVec &v = Lvecs[q];
                bool vb = v.n < 2 || v.n > 176 || v.x < 0 || v.x + v.n > 2 * N || v.recombine > 3 || v.time_divide > 3 || v.b_tree < 1 ||
                          v.b_tree > 16 || v.b_in < 1 || v.b_in > 8 || v.leaf0 < 0 || v.leaf1 < v.leaf0 || v.leaf1 - v.leaf0 > 16 ||
                          v.leaf1 > nleaves || v.band > 20 || v.fill_lo > v.fill_hi || v.fill_hi > 21 || v.fill_mode > 3 || v.sel > 1 ||
                          v.fold + v.n > kShapeNorm / 2 || v.out + v.n > kShapeNorm / 2 || (int)v.nb_tree * (int)v.b_tree != (int)v.n ||
                          v.n % v.b_in != 0 || (v.b_in >> v.recombine) < 1 || ((int)v.b_tree << v.recombine) > 16;
                if (!vb)
                    for (int l = v.leaf0; l < v.leaf1; l++) {        // its leaves lie inside it
                        const SymLeaf &lf = Lleaves[l];
                        vb |= lf.off + lf.n > v.n || lf.fold_off + lf.n > v.n || lf.abs != v.x + lf.off;
                    }
                bad |= vb;
            }
            for (int q = lane; q < nops; q += kWave) {
                const SymOp &o = Lops[q];
                bool ob;
                switch (o.kind) {
                case 0: ob = o.a < 0 || o.a >= nvecs; break;
                case 1: ob = o.a < 0 || o.a >= 2 * N || o.b >= kShapeNorm / 2 || o.band > 20; break;
                case 2: ob = o.a < 0 || o.b < 0 || o.a + 2 > 2 * N || o.b + 2 > 2 * N; break;
                case 3: ob = o.a < 0 || o.b < 0 || o.n < 0 || o.a + o.n > 2 * N || o.b + o.n > 2 * N; break;
                case 4: ob = o.a < 0 || o.n < 0 || o.a + o.n > 2 * N; break;
                case 5: ob = o.a < 0 || o.a > kShapeNorm / 2; break;
                default: ob = true; break;
                }
                bad |= ob;
            }
            if (__any(bad)) {
                float4 *o4 = reinterpret_cast<float4 *>(out);
                for (int j = lane; j < CC * N / 4; j += kWave) o4[j] = float4{0.f, 0.f, 0.f, 0.f};
                continue;
            }
        }
        // pass A: every pulse leaf of the frame, a leaf per lane
        for (int l = lane; l < nleaves && !NYQ_SHAPE_DBG_NO_A; l += kWave) {
            const SymLeaf &lf = Lleaves[l];
            if (lf.kind != 0) continue;
            unsigned cm = 0;
            lane_pulse_leaf(tab, lf, X, H.spread, &cm);
            L.leafCm[l] = (unsigned short)cm;
        }
        NYQ_WAVE_SYNC();
        // pass B: the operation list
        for (int q = 0; q < nops && !NYQ_SHAPE_DBG_NO_B; q++) {
            const SymOp o = F.ops[q];
            switch (o.kind) {
            case 0:
                shape_vector(F, F.vecs[o.a], X, norm, norm2, work, tmp, seed, lane);
                break;
            case 1:
                if (lane == 0) {
                    X[o.a] = o.f0;
                    if (o.b >= 0) (o.n ? norm2 : norm)[o.b] = o.f0;
                    L.masks[o.band] |= 1;
                    L.masks[21 + o.band] |= 1;
                }
                NYQ_WAVE_SYNC();
                break;
            case 2:
                if (lane == 0) {
                    float *x = X + o.a, *y = X + o.b;
                    const int sign = (o.n & 1) ? -1 : 1;
                    float *x2 = (o.n & 2) ? y : x, *y2 = (o.n & 2) ? x : y;
                    y2[0] = -sign * x2[1];
                    y2[1] = sign * x2[0];
                    const float x0 = o.f0 * x[0], x1 = o.f0 * x[1], y0 = o.f1 * y[0], y1 = o.f1 * y[1];
                    x[0] = x0 - y0;
                    y[0] = x0 + y0;
                    x[1] = x1 - y1;
                    y[1] = x1 + y1;
                }
                NYQ_WAVE_SYNC();
                break;
            case 3: {                                                // mid / side -> left / right (bands.c:391-441)
                float *x = X + o.a, *y = X + o.b;
                const float mid = o.f0;
                float a = 0.f, b = 0.f;
                for (int j = lane; j < o.n; j += kWave) {
                    a += y[j] * x[j];
                    b += y[j] * y[j];
                }
                const float xp = mid * wave_sum(a), side = wave_sum(b);
                const float el = mid * mid + side - 2 * xp, er = mid * mid + side + 2 * xp;
                if (er < 6e-4f || el < 6e-4f) {
                    for (int j = lane; j < o.n; j += kWave) y[j] = x[j];
                } else {
                    const float lg = 1.f / sqrtf(el), rg = 1.f / sqrtf(er);
                    for (int j = lane; j < o.n; j += kWave) {
                        const float l = mid * x[j], rr = y[j];
                        x[j] = lg * (l - rr);
                        y[j] = rg * (l + rr);
                    }
                }
                NYQ_WAVE_SYNC();
                break;
            }
            case 4:
                for (int j = lane; j < o.n; j += kWave) X[o.a + j] = -X[o.a + j];
                NYQ_WAVE_SYNC();
                break;
            default:
                for (int j = lane; j < o.a; j += kWave) norm[j] = .5f * (norm[j] + norm2[j]);
                NYQ_WAVE_SYNC();
                break;
            }
        }
        // anti-collapse (bands.c:258-351): short blocks of a transient frame that received nothing get noise at the level the
        // host computed from the energies, then the band is renormalised; the generator goes on from where the fills left it
        if (H.flags & 2) {
            const float *level = reinterpret_cast<const float *>(r + levelOff);
            const float laneLevel = lane < 2 * 21 ? level[lane] : 0.f;
            for (int i = H.start; i < H.end; i++) {
                const int e0 = shape_edge(i, LM), n0 = (shape_edge(i + 1, LM) - e0) >> LM, M = 1 << LM;
                for (int c = 0; c < C; c++) {
                    const unsigned m = L.masks[c * 21 + i];
                    if ((m & ((1u << M) - 1)) == (1u << M) - 1) continue;           // every block of the band received something
                    const float rl = __shfl(laneLevel, c * 21 + i);
                    float *x = X + c * N + e0;
                    for (int k = 0; k < M; k++) {
                        if (m >> k & 1) continue;
                        for (int j = lane; j < n0; j += kWave) x[(j << LM) + k] = (lcg_jump(seed, j + 1) & 0x8000u) ? rl : -rl;
                        seed = lcg_jump(seed, n0);
                    }
                    NYQ_WAVE_SYNC();
                    float e = 0.f;
                    for (int j = lane; j < M * n0; j += kWave) e += x[j] * x[j];
                    const float g = 1.f / sqrtf(wave_sum(e) + 1e-15f);
                    for (int j = lane; j < M * n0; j += kWave) x[j] = g * x[j];
                    NYQ_WAVE_SYNC();
                }
            }
        }
        // denormalise_bands: every band times its gain, zeros below `start` and above `end`; a packet that codes one channel
        // of a stereo stream is played on both, one that codes two for a mono stream is mixed down
        const int lo = shape_edge(H.start, LM), hi = shape_edge(H.end, LM);
        for (int c = 0; c < CC; c++) {
            float *fo = out + c * N;
            for (int j = lane; j < lo; j += kWave) fo[j] = 0.f;
            for (int j = hi + lane; j < N; j += kWave) fo[j] = 0.f;
        }
        for (int i = H.start; i < H.end; i++) {
            const int e0 = shape_edge(i, LM), e1 = shape_edge(i + 1, LM);
            const float g0 = __shfl(laneGain, i), g1 = __shfl(laneGain, 21 + i);
            for (int j = e0 + lane; j < e1; j += kWave) {
                const float a = X[j] * g0;
                if (C == CC) {
                    out[j] = a;
                    if (C == 2) out[N + j] = X[N + j] * g1;
                } else if (C == 1) {
                    out[j] = a;
                    out[N + j] = a;
                } else {
                    out[j] = .5f * (a + X[N + j] * g1);
                }
            }
        }
    }
}

}  // namespace nyq
