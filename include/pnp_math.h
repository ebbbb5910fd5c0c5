/* pnp_math.h -- the numerical contract for the transcendental steps of the post-process path.
 *
 * softmax / -log(clip(p)) / mean-field exp-and-normalise (reference call sites:
 * PnP_OVSS_0514_updated_segmentation.py:1054-1071 -> torch F.softmax, pydensecrf
 * unary_from_softmax, DenseCRF::inference) are defined HERE as fixed fmaf sequences so that the
 * HIP kernels and the CPU oracle evaluate bit-identical values (libm / ocml / Eigen / Sleef all
 * differ from each other in the last ulp, which would make "argmax label maps bit-exact"
 * untestable).  Accuracy: < 1.5 ulp on the ranges used (exp: [-88, 88], log: [1e-5, 1]).
 * No a*b+c expression is left for the compiler to contract: every multiply-add is an explicit fmaf.
 */
#ifndef PNP_MATH_H
#define PNP_MATH_H

#include <math.h>
#include <stdint.h>

#if defined(__HIPCC__)
#define PNP_HD __host__ __device__ __forceinline__
#else
#define PNP_HD static inline
#endif

PNP_HD float pnp_bits_to_float(uint32_t u) {
    union { float f; uint32_t u; } v;
    v.u = u;
    return v.f;
}

PNP_HD uint32_t pnp_float_to_bits(float f) {
    union { float f; uint32_t u; } v;
    v.f = f;
    return v.u;
}

/* exp(x), round-to-nearest-even argument reduction, degree-6 minimax (Cephes coefficients). */
PNP_HD float pnp_expf(float x) {
    if (!(x == x)) return x;
    if (x > 88.72283f) return pnp_bits_to_float(0x7f800000u);
    if (x < -87.33654f) return 0.0f;
    float n = rintf(x * 1.44269504f);
    float r = fmaf(n, -0.693359375f, x);
    r = fmaf(n, 2.12194440e-4f, r);
    float p = 1.9875691500e-4f;
    p = fmaf(p, r, 1.3981999507e-3f);
    p = fmaf(p, r, 8.3334519073e-3f);
    p = fmaf(p, r, 4.1665795894e-2f);
    p = fmaf(p, r, 1.6666665459e-1f);
    p = fmaf(p, r, 5.0000001201e-1f);
    float r2 = r * r;
    float y = fmaf(p, r2, r);
    y = y + 1.0f;
    int ni = (int)n;
    if (ni > 127) {
        y = y * 2.0f;
        ni = 127;
    }
    return y * pnp_bits_to_float((uint32_t)(ni + 127) << 23);
}

/* log(x) for finite normal x > 0 (Cephes logf structure). */
PNP_HD float pnp_logf(float x) {
    if (!(x == x)) return x;
    uint32_t u = pnp_float_to_bits(x);
    int e = (int)((u >> 23) & 0xffu) - 126;
    float m = pnp_bits_to_float((u & 0x007fffffu) | 0x3f000000u);
    if (m < 0.70710678f) {
        e -= 1;
        m = m + m;
    }
    m = m - 1.0f;
    float z = m * m;
    float p = 7.0376836292e-2f;
    p = fmaf(p, m, -1.1514610310e-1f);
    p = fmaf(p, m, 1.1676998740e-1f);
    p = fmaf(p, m, -1.2420140846e-1f);
    p = fmaf(p, m, 1.4249322787e-1f);
    p = fmaf(p, m, -1.6668057665e-1f);
    p = fmaf(p, m, 2.0000714765e-1f);
    p = fmaf(p, m, -2.4999993993e-1f);
    p = fmaf(p, m, 3.3333331174e-1f);
    float y = (p * m) * z;
    float fe = (float)e;
    y = fmaf(fe, -2.12194440e-4f, y);
    y = fmaf(z, -0.5f, y);
    float r = m + y;
    r = fmaf(fe, 0.693359375f, r);
    return r;
}

#endif /* PNP_MATH_H */
