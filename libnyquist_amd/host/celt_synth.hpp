// celt_synth.hpp -- the FLOAT half of a CELT frame's band shapes, as flat kernels over whole vectors.
//
// The entropy stage (celt_decoder.cpp) first reads a band's symbols into a small list of leaves (integer pulse vectors,
// fill decisions, gains) and only then builds the band's normalised coefficients with the kernels below -- nothing in
// here touches the bitstream.  Every reduction runs over EIGHT interleaved partial sums combined in a fixed order: the
// compiler keeps them in vector registers and the value does not depend on the vector width it picks; element-wise
// kernels need nothing special to vectorise.  (Cloning the kernels for AVX2 behind an ifunc was measured: the vectors are
// 8 .. 176 floats and the indirect calls cost what the wider registers gain.)
// What the kernels compute is what RFC 6716 section 4.3.4 prescribes (reference: vq.c, bands.c of libopus); how the
// sums are ordered is this decoder's own (results agree with the reference's scalar loops to float rounding).
#pragma once
#include <cmath>
#include <cstdint>
#include <cstring>

#define NYQ_CLONES

namespace nyq_host {
namespace synth {

// sum of x[i]^2 (+ init), eight interleaved partial sums
NYQ_CLONES inline float energy(const float *x, int n, float init) {
    float acc[8] = {init, 0, 0, 0, 0, 0, 0, 0};
    int i = 0;
    for (; i + 8 <= n; i += 8)
        for (int l = 0; l < 8; l++) acc[l] += x[i + l] * x[i + l];
    for (int l = 0; i < n; i++, l++) acc[l] += x[i] * x[i];
    return ((acc[0] + acc[4]) + (acc[2] + acc[6])) + ((acc[1] + acc[5]) + (acc[3] + acc[7]));
}

// sum of x[i] * y[i] and of y[i]^2 in one pass (stereo merge)
NYQ_CLONES inline void crossEnergy(const float *x, const float *y, int n, float &xy, float &yy) {
    float a[8] = {0, 0, 0, 0, 0, 0, 0, 0}, b[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    int i = 0;
    for (; i + 8 <= n; i += 8)
        for (int l = 0; l < 8; l++) {
            a[l] += y[i + l] * x[i + l];
            b[l] += y[i + l] * y[i + l];
        }
    for (int l = 0; i < n; i++, l++) {
        a[l] += y[i] * x[i];
        b[l] += y[i] * y[i];
    }
    xy = ((a[0] + a[4]) + (a[2] + a[6])) + ((a[1] + a[5]) + (a[3] + a[7]));
    yy = ((b[0] + b[4]) + (b[2] + b[6])) + ((b[1] + b[5]) + (b[3] + b[7]));
}

NYQ_CLONES inline void scale(float *x, int n, float g) {
    for (int i = 0; i < n; i++) x[i] = g * x[i];
}
NYQ_CLONES inline void scaleTo(float *dst, const float *src, int n, float g) {
    for (int i = 0; i < n; i++) dst[i] = g * src[i];
}
NYQ_CLONES inline void negate(float *x, int n) {
    for (int i = 0; i < n; i++) x[i] = -x[i];
}

// a pulse vector as unit-norm coefficients times `gain`: x = y * gain / |y|  (yy = |y|^2, an exact small integer)
inline void fromPulses(float *x, const int16_t *y, int n, float gain, int32_t yy) {
    const float g = (1.f / std::sqrt((float)yy)) * gain;
    for (int j = 0; j < n; j++) x[j] = g * (float)y[j];
}

// unit norm times gain (vq.c renormalise_vector)
inline void renormalise(float *x, int n, float gain) {
    const float g = (1.f / std::sqrt(energy(x, n, 1e-15f))) * gain;
    scale(x, n, g);
}

// One level of the Haar transform over `stride` interleaved sequences (bands.c haar1): pairs (2j, 2j+1) of each.
NYQ_CLONES inline void haar(float *x, int n0, int stride) {
    const float r = .70710678f;
    const int half = n0 >> 1;
    if (stride == 1) {
        for (int j = 0; j < half; j++) {
            const float a = r * x[2 * j], b = r * x[2 * j + 1];
            x[2 * j] = a + b;
            x[2 * j + 1] = a - b;
        }
        return;
    }
    for (int j = 0; j < half; j++) {
        float *p = x + stride * 2 * j, *q = p + stride;
        for (int i = 0; i < stride; i++) {
            const float a = r * p[i], b = r * q[i];
            p[i] = a + b;
            q[i] = a - b;
        }
    }
}

// mid / side back to left / right with the energies equalised (bands.c stereo_merge)
NYQ_CLONES inline void stereoMerge(float *x, float *y, float mid, int n) {
    float xp, side;
    crossEnergy(x, y, n, xp, side);
    xp = mid * xp;
    const float el = mid * mid + side - 2 * xp, er = mid * mid + side + 2 * xp;
    if (er < 6e-4f || el < 6e-4f) {
        std::memcpy(y, x, sizeof(float) * (size_t)n);
        return;
    }
    const float lg = 1.f / std::sqrt(el), rg = 1.f / std::sqrt(er);
    for (int j = 0; j < n; j++) {
        const float l = mid * x[j], r = y[j];
        x[j] = lg * (l - r);
        y[j] = rg * (l + r);
    }
}

// The spreading rotation (vq.c exp_rotation1) on `nch` vectors of one length at once: a pass is a first-order recurrence
// along the vector -- every step waits for the one before it -- so vectors that share (len, stride) are walked in lockstep
// and their chains fill each other's latency.  c[k], s[k]: the rotation of vector k.
template <int NCH>
inline void rotatePass(float *const *x, const float *c, const float *s, int len, int stride) {
    for (int i = 0; i < len - stride; i++)
        for (int k = 0; k < NCH; k++) {
            float *p = x[k] + i;
            const float x1 = p[0], x2 = p[stride];
            p[stride] = c[k] * x2 + s[k] * x1;
            p[0] = c[k] * x1 - s[k] * x2;
        }
    for (int i = len - 2 * stride - 1; i >= 0; i--)
        for (int k = 0; k < NCH; k++) {
            float *p = x[k] + i;
            const float x1 = p[0], x2 = p[stride];
            p[stride] = c[k] * x2 + s[k] * x1;
            p[0] = c[k] * x1 - s[k] * x2;
        }
}
inline void rotateChains(float *const *x, const float *c, const float *s, int nch, int len, int stride) {
    int k = 0;
    for (; k + 4 <= nch; k += 4) rotatePass<4>(x + k, c + k, s + k, len, stride);
    if (k + 2 <= nch) { rotatePass<2>(x + k, c + k, s + k, len, stride); k += 2; }
    if (k < nch) rotatePass<1>(x + k, c + k, s + k, len, stride);
}

}  // namespace synth
}  // namespace nyq_host
