/*
 * rectipy_math.h -- bit-reproducible fp32 transcendental functions.
 *
 * The activation functions of the rate-neuron templates
 * (reference: neuron_model_templates/rate_neurons/leaky_integrator.yaml:22 sigmoid,
 *  :33 tanh) need expf/tanhf.  The device libm (ocml) and the host libm (glibc)
 * round differently in the last ulp, which would make HIP-vs-CPU parity a
 * tolerance test and hide indexing bugs.  These versions use only IEEE-754
 * correctly rounded primitives (fma, mul, add, div, rint) in a fixed order, so
 * gcc on the host and hipcc on gfx950 produce the same bits.  Both sides must be
 * compiled with -ffp-contract=off (the code spells every fma explicitly).
 *
 * Accuracy (checked in tests/test_math.py against fp64 libm): <= 2 ulp.
 */
#ifndef RECTIPY_MATH_H
#define RECTIPY_MATH_H

#include <math.h>
#include <stdint.h>

#if defined(__HIPCC__)
#define RP_HD __host__ __device__ __forceinline__
#else
#define RP_HD static inline
#endif

RP_HD float rp_pow2i(int n) { /* 2^n for n in [-126, 127], exact */
    uint32_t bits = (uint32_t)(n + 127) << 23;
    float f;
    __builtin_memcpy(&f, &bits, sizeof f);
    return f;
}

/* e^x, Cody-Waite reduction + degree-5 minimax polynomial (Cephes coefficients). */
RP_HD float rp_expf(float x) {
    if (x != x) return x;
    if (x > 88.3762626647949f) return INFINITY;
    if (x < -87.3365478515625f) return 0.0f; /* flushes the subnormal range */
    float n = rintf(x * 1.44269504088896341f);
    float r = fmaf(n, -0.693359375f, x);
    r = fmaf(n, 2.12194440e-4f, r);
    float z = r * r;
    float p = 1.9875691500e-4f;
    p = fmaf(p, r, 1.3981999507e-3f);
    p = fmaf(p, r, 8.3334519073e-3f);
    p = fmaf(p, r, 4.1665795894e-2f);
    p = fmaf(p, r, 1.6666665459e-1f);
    p = fmaf(p, r, 5.0000001201e-1f);
    float y = fmaf(p, z, r) + 1.0f;
    int ni = (int)n;
    int h = ni >> 1; /* two exact scalings keep 2^n representable at both ends */
    return (y * rp_pow2i(h)) * rp_pow2i(ni - h);
}

/* tanh(x): odd polynomial near 0, 1 - 2/(e^{2|x|}+1) elsewhere. */
RP_HD float rp_tanhf(float x) {
    float ax = fabsf(x);
    if (ax < 0.625f) {
        float z = x * x;
        float p = -5.70498872745e-3f;
        p = fmaf(p, z, 2.06390887954e-2f);
        p = fmaf(p, z, -5.37397155531e-2f);
        p = fmaf(p, z, 1.33314422036e-1f);
        p = fmaf(p, z, -3.33332819422e-1f);
        return fmaf(p * z, x, x);
    }
    if (ax > 10.0f) return copysignf(1.0f, x); /* also +-inf */
    if (x != x) return x;
    float e = rp_expf(ax + ax);
    float t = 1.0f - 2.0f / (e + 1.0f);
    return copysignf(t, x);
}

#endif /* RECTIPY_MATH_H */
