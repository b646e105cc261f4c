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
// One wavefront = one frame at a time (grid-stride over frames); the frame's coefficients X[C][960] and the fold memory live
// in the wave's LDS slice; the control flow is the record list, wave-uniform.  Rotation passes are chains of dependent steps:
// the residues of a pass (and the interleaved short blocks) are independent chains and run on different lanes, each with the
// carried value in a register.
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
constexpr int kSymN = 960;                                           // bins per channel (LM 3)
constexpr int kSymFixed = 32 + 42 * 4 + kSymMaxOps * 16 + kSymMaxVecs * 24;   // 3064
constexpr int kSymBodyOff = (kSymFixed + 15) & ~15;                  // 3072
constexpr int kSymMaxLeaves = 192;                                   // 96 per channel: what the body holds
constexpr int kPvqDim = 178;                                         // U(n, k) for n, k < 178 (the widest band has 176 bins)
__host__ __device__ inline size_t sym_bytes(int channels) { return (size_t)kSymBodyOff + (size_t)channels * kSymN * 4; }

constexpr int kShapeNorm = 2 * 800;                                  // fold memory: two channels x bins below the last band
constexpr int kShapeLdsFloats = 2 * kSymN + kShapeNorm + 192 + 192;

// U(n, k) of cwrs.c (the number of k-pulse vectors in n dimensions whose first coordinate is not negative ... ), 32-bit,
// saturated: entries a valid codeword never reaches.  Built once per context on the host (shape_core).
inline void pvq_table_build(unsigned *T) {
    const unsigned long long cap = ~0ull >> 1;
    static unsigned long long U[kPvqDim][kPvqDim];
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
    for (int n = 0; n < kPvqDim; n++)
        for (int k = 0; k < kPvqDim; k++) T[n * kPvqDim + k] = U[n][k] > 0xFFFFFFFFull ? 0xFFFFFFFFu : (unsigned)U[n][k];
}

// codeword -> pulse vector y[0 .. n) (cwrs.c cwrsi, 32-bit rows): one lane, its leaf.  Returns |y|^2; *cm = which of the
// leaf's `blocks` interleaved short blocks received a pulse.
__device__ __forceinline__ int pvq_unrank(const unsigned *__restrict__ T, int n, int k, unsigned idx, short *y, int blocks, unsigned *cm) {
    const int per = blocks > 1 ? n / blocks : n;
    int yy = 0, j = 0;
    unsigned mask = 0;
    while (n > 2) {
        unsigned a = T[(k + 1) * kPvqDim + n];                       // U(n, k + 1)
        const bool neg = idx >= a;
        if (neg) idx -= a;
        a = T[k * kPvqDim + n];
        int v = 0;
        if (a <= idx) {
            idx -= a;
        } else {
            int kk = k;
            if (kk > n && T[n * kPvqDim + n] > idx) kk = n;
            unsigned p;
            do {
                kk--;
                p = T[kk * kPvqDim + n];
            } while (p > idx);
            idx -= p;
            v = k - kk;
            k = kk;
        }
        y[j] = (short)(neg ? -v : v);
        yy += v * v;
        if (v) mask |= 1u << (j / per);
        j++;
        n--;
    }
    {
        const unsigned a = 2 * (unsigned)k + 1;
        const bool neg = idx >= a;
        if (neg) idx -= a;
        const int kk = (int)((idx + 1) >> 1);
        if (kk) idx -= 2 * (unsigned)kk - 1;
        const int v = k - kk;
        y[j] = (short)(neg ? -v : v);
        y[j + 1] = (short)(idx ? -kk : kk);
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
__device__ __forceinline__ int wave_sum_i(int v) {
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

// one rotation pass (vq.c exp_rotation1) over `blocks` consecutive blocks of `len` floats at x: the residues modulo `stride`
// of a block are independent chains; a lane owns one (block, residue), forward walk then backward walk, the carried value in
// a register
__device__ __forceinline__ void shape_rotate(float *x, int blocks, int len, int stride, float c, float s, int lane) {
#pragma clang fp contract(off)
    const int chains = blocks * stride;
    for (int t = lane; t < chains; t += kWave) {
        const int blk = t / stride, r = t - blk * stride;
        float *p = x + blk * len;
        // forward: i = r, r + stride, ... < len - stride
        if (r < len - stride) {
            float a = p[r];
            int i = r;
            for (; i < len - stride; i += stride) {
                const float x2 = p[i + stride];
                const float hi = c * x2 + s * a;
                p[i] = c * a - s * x2;
                a = hi;
            }
            p[i] = a;
        }
        // backward: i = the largest index = r (mod stride) that is <= len - 2 stride - 1, down to r
        const int top = len - 2 * stride - 1;
        if (r <= top) {
            int i = top - ((top - r) % stride);
            float b = p[i + stride];
            for (; i >= 0; i -= stride) {
                const float x1 = p[i];
                p[i + stride] = c * b + s * x1;
                b = c * x1 - s * b;
            }
            p[i + stride] = b;
        }
    }
    NYQ_WAVE_SYNC();
}

struct ShapeFrame {
    const SymHead *head;
    const float *gain;
    const SymOp *ops;
    const SymVec *vecs;
    const SymLeaf *leaves;
    const short *pulses;             // LDS: the frame's pulse vectors at the offsets of their coefficients
    const int *energy;               // LDS: |y|^2 per leaf
    const unsigned short *leafCm;    // LDS: collapse mask of a pulse leaf
    unsigned char *masks;            // LDS: [2][21] collapse masks of the bands so far
};

__device__ __forceinline__ void shape_vector(const ShapeFrame &F, const SymVec v, float *X, float *norm, float *norm2, float *work,
                                             float *tmp, unsigned &seed, int spread, int lane) {
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
        const SymLeaf lf = F.leaves[l];
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
    // leaves in tree order (the noise generator advances through the filled ones in this order)
    for (int l = v.leaf0; l < v.leaf1; l++) {
        const SymLeaf lf = F.leaves[l];
        const unsigned kind = kinds >> (2 * (l - v.leaf0)) & 3;
        float *o = x + lf.off;
        const int ln = lf.n;
        if (kind == 0) {
            const short *y = F.pulses + lf.abs;
            const float g = (1.f / sqrtf((float)F.energy[l])) * lf.gain;
            for (int j = lane; j < ln; j += kWave) o[j] = g * (float)y[j];
            NYQ_WAVE_SYNC();
            if (spread != 0 && 2 * lf.k < ln) {
                const int factor = spread == 1 ? 15 : spread == 2 ? 10 : 5;
                const float gn = (float)(1.0f * ln) / (float)(ln + factor * lf.k);
                const float theta = .5f * (gn * gn);
                const float c = (float)cos((double)((.5f * 3.141592653f) * theta));
                const float s = (float)cos((double)((.5f * 3.141592653f) * (1.0f - theta)));
                const int stride = lf.blocks, len = ln / stride;
                int stride2 = 0;
                if (ln >= 8 * stride) {
                    stride2 = 1;
                    while ((stride2 * stride2 + stride2) * stride + (stride >> 2) < ln) stride2++;
                }
                if (stride2) shape_rotate(o, stride, len, stride2, s, c, lane);
                shape_rotate(o, stride, len, 1, c, s, lane);
            }
        } else if (kind == 1) {
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

// band edges of the 48 kHz mode at 20 ms (modes.c:41-44 times 8)
__device__ __forceinline__ int shape_edge(int i) {
    const short e[22] = {0, 8, 16, 24, 32, 40, 48, 56, 64, 80, 96, 112, 128, 160, 192, 224, 272, 320, 384, 480, 624, 800};
    return e[i];
}

__global__ __launch_bounds__(kWave) void celt_shape_kernel(const unsigned char *__restrict__ sym, float *__restrict__ freq,
                                                           const unsigned *__restrict__ pvq, long nstreams, long nframes, int channels,
                                                           long sstride) {
#pragma clang fp contract(off)
    __shared__ __attribute__((aligned(16))) float lds[kShapeLdsFloats];
    __shared__ short pulses[2 * kSymN];
    __shared__ int energy[kSymMaxLeaves];
    __shared__ unsigned short leafCm[kSymMaxLeaves];
    __shared__ unsigned char masks[2 * 21 + 6];
    float *X = lds, *norm = lds + 2 * kSymN, *work = norm + kShapeNorm, *tmp = work + 192;
    const int lane = threadIdx.x;
    const size_t rec = sym_bytes(channels);
    const long total = nstreams * nframes;
    for (long u = blockIdx.x; u < total; u += gridDim.x) {
        const long s = u / nframes, f = u - s * nframes;
        const unsigned char *r = sym + ((size_t)s * (size_t)sstride + (size_t)f) * rec;
        float *out = freq + (size_t)u * (size_t)channels * kSymN;
        const SymHead H = *reinterpret_cast<const SymHead *>(r);
        const int C = channels;
        if (H.flags & 1) {                                           // the host built this frame itself: its freq[] is the body
            const float4 *b4 = reinterpret_cast<const float4 *>(r + kSymBodyOff);
            float4 *o4 = reinterpret_cast<float4 *>(out);
            for (int j = lane; j < C * kSymN / 4; j += kWave) o4[j] = b4[j];
            continue;
        }
        if (H.nops == 0) {                                           // a record of zeros: a silent (or padding) frame
            float4 *o4 = reinterpret_cast<float4 *>(out);
            for (int j = lane; j < C * kSymN / 4; j += kWave) o4[j] = float4{0.f, 0.f, 0.f, 0.f};
            continue;
        }
        ShapeFrame F;
        F.head = reinterpret_cast<const SymHead *>(r);
        F.gain = reinterpret_cast<const float *>(r + 32);
        F.ops = reinterpret_cast<const SymOp *>(r + 32 + 42 * 4);
        F.vecs = reinterpret_cast<const SymVec *>(r + 32 + 42 * 4 + kSymMaxOps * 16);
        F.leaves = reinterpret_cast<const SymLeaf *>(r + kSymBodyOff);
        F.pulses = pulses;
        F.energy = energy;
        F.leafCm = leafCm;
        F.masks = masks;
        float *norm2 = norm + (shape_edge(20) - shape_edge(H.start));
        unsigned seed = H.seed;
        NYQ_WAVE_SYNC();
        // (a) every pulse vector of the frame from its codeword: a leaf per lane
        if (lane < 2 * 21) masks[lane] = 0;
        const int nleaves = H.nleaves < kSymMaxLeaves ? H.nleaves : kSymMaxLeaves;
        for (int l = lane; l < nleaves; l += kWave) {
            const SymLeaf lf = F.leaves[l];
            if (lf.kind != 0) continue;
            unsigned cm;
            energy[l] = pvq_unrank(pvq, lf.n, lf.k, lf.index, pulses + lf.abs, lf.blocks, &cm);
            leafCm[l] = (unsigned short)cm;
        }
        NYQ_WAVE_SYNC();
        for (int q = 0; q < H.nops; q++) {
            const SymOp o = F.ops[q];
            switch (o.kind) {
            case 0: shape_vector(F, F.vecs[o.a], X, norm, norm2, work, tmp, seed, H.spread, lane); break;
            case 1:
                if (lane == 0) {
                    X[o.a] = o.f0;
                    if (o.b >= 0) (o.n ? norm2 : norm)[o.b] = o.f0;
                    masks[o.band] |= 1;
                    masks[21 + o.band] |= 1;
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
        // denormalise_bands: every band times its gain, zeros below `start` and above `end`
        for (int c = 0; c < C; c++) {
            float *fo = out + c * kSymN;
            const float *x = X + c * kSymN;
            const int lo = shape_edge(H.start), hi = shape_edge(H.end);
            for (int j = lane; j < lo; j += kWave) fo[j] = 0.f;
            for (int i = H.start; i < H.end; i++) {
                const float g = F.gain[c * 21 + i];
                const int e0 = shape_edge(i), e1 = shape_edge(i + 1);
                for (int j = e0 + lane; j < e1; j += kWave) fo[j] = x[j] * g;
            }
            for (int j = hi + lane; j < kSymN; j += kWave) fo[j] = 0.f;
        }
        NYQ_WAVE_SYNC();
    }
}

}  // namespace nyq
