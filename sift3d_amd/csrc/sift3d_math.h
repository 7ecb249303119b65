/* sift3d_math.h -- scalar math shared by the device kernels and by host-side
 * self-checks (compiled by hipcc for gfx950 and by gcc for the host shim).
 *
 * Why this exists: parity with the reference's CPU path is judged on keypoint
 * COUNTS (threshold decisions) and 1e-5-relative floats.  The reference calls
 * glibc's expf() per window voxel (sift.c:972, sift.c:1498) and LAPACK for a
 * 3x3 symmetric eigen-problem (imutil.c:1027-1045).  Neither exists on the
 * device, so both are provided here in a form whose results are reproducible
 * across host and device:
 *
 *  s3d_expf()   the algorithm glibc >= 2.27 uses for expf (Szabolcs Nagy's
 *               exp2f-table method: 32-entry 2^(i/32) table, cubic in double),
 *               restated from its published description.  tests/ check it is
 *               bit-identical to the host libm over tens of millions of
 *               arguments in the range the windows use.
 *  s3d_eigen3() cyclic Jacobi in double (eigenvalues ascending, eigenvectors in
 *               columns); the consumer removes the sign ambiguity itself
 *               (sift.c:1038-1042).
 *
 * All code here must be compiled with FP contraction OFF.
 */
#ifndef SIFT3D_AMD_MATH_H
#define SIFT3D_AMD_MATH_H

#include <math.h>
#include <stdint.h>
#include <string.h>

#if defined(__HIPCC__)
#define S3D_HD __host__ __device__ __forceinline__
#else
#define S3D_HD static inline
#endif

/* T[i] = bits(2^(i/32)) - (i << 47) */
#if defined(__HIPCC__) && defined(__HIP_DEVICE_COMPILE__)
__device__ __constant__
#else
static const
#endif
uint64_t s3d_exp2_tab[32] = {
    0x3ff0000000000000ull, 0x3fefd9b0d3158574ull, 0x3fefb5586cf9890full, 0x3fef9301d0125b51ull,
    0x3fef72b83c7d517bull, 0x3fef54873168b9aaull, 0x3fef387a6e756238ull, 0x3fef1e9df51fdee1ull,
    0x3fef06fe0a31b715ull, 0x3feef1a7373aa9cbull, 0x3feedea64c123422ull, 0x3feece086061892dull,
    0x3feebfdad5362a27ull, 0x3feeb42b569d4f82ull, 0x3feeab07dd485429ull, 0x3feea47eb03a5585ull,
    0x3feea09e667f3bcdull, 0x3fee9f75e8ec5f74ull, 0x3feea11473eb0187ull, 0x3feea589994cce13ull,
    0x3feeace5422aa0dbull, 0x3feeb737b0cdc5e5ull, 0x3feec49182a3f090ull, 0x3feed503b23e255dull,
    0x3feee89f995ad3adull, 0x3feeff76f2fb5e47ull, 0x3fef199bdd85529cull, 0x3fef3720dcef9069ull,
    0x3fef5818dcfba487ull, 0x3fef7c97337b9b5full, 0x3fefa4afa2a490daull, 0x3fefd0765b6e4540ull,
};

S3D_HD double s3d_u64_as_f64(uint64_t u)
{
    double d;
    memcpy(&d, &u, sizeof(d));
    return d;
}

S3D_HD uint64_t s3d_f64_as_u64(double d)
{
    uint64_t u;
    memcpy(&u, &d, sizeof(u));
    return u;
}

/* expf.  Arguments in [-150, 88.72] take the table path (denormal results are
 * produced by the final double -> float conversion, as in glibc); outside it the
 * limits (0 / +inf) are returned without libm's errno/fenv side effects.
 * `use_fma` selects the contraction pattern of glibc's FMA-enabled build
 * (what an x86-64 host with FMA3 dispatches to). */
/* the table path itself: valid for -150 <= x <= 88.72 (branch-free) */
S3D_HD float s3d_expf_core(float x, int use_fma, const uint64_t *tab)
{
    const double inv_ln2_n = 0x1.71547652b82fep+0 * 32.0;
    const double shift = 0x1.8p+52;
    const double c0 = 0x1.c6af84b912394p-5 / 32.0 / 32.0 / 32.0;
    const double c1 = 0x1.ebfce50fac4f3p-3 / 32.0 / 32.0;
    const double c2 = 0x1.62e42ff0c52d6p-1 / 32.0;
    double xd, z, kd, r, r2, y, s;
    uint64_t ki, t;

    xd = (double)x;
    z = inv_ln2_n * xd;
    kd = z + shift;
    ki = s3d_f64_as_u64(kd);
    kd -= shift;
    r = z - kd;
    t = tab[ki % 32];
    t += ki << (52 - 5);
    s = s3d_u64_as_f64(t);
    if (use_fma) {
        z = fma(c0, r, c1);
        r2 = r * r;
        y = fma(c2, r, 1.0);
        y = fma(z, r2, y);
    } else {
        z = c0 * r + c1;
        r2 = r * r;
        y = c2 * r + 1.0;
        y = z * r2 + y;
    }
    y = y * s;
    return (float)y;
}

S3D_HD float s3d_expf_tab(float x, int use_fma, const uint64_t *tab)
{
    if (!(x >= -150.0f))
        return x != x ? x : 0.0f;
    if (x > 0x1.62e42ep6f)
        return INFINITY;
    return s3d_expf_core(x, use_fma, tab);
}

S3D_HD float s3d_expf_impl(float x, int use_fma) { return s3d_expf_tab(x, use_fma, s3d_exp2_tab); }

#ifndef S3D_EXPF_FMA
#define S3D_EXPF_FMA 1
#endif

S3D_HD float s3d_expf(float x) { return s3d_expf_impl(x, S3D_EXPF_FMA); }

/* same, with the caller's copy of s3d_exp2_tab (the kernels keep one in LDS) */
S3D_HD float s3d_expf_with(float x, const uint64_t *tab) { return s3d_expf_tab(x, S3D_EXPF_FMA, tab); }
/* the caller guarantees -150 <= x <= 88.72: no range branches */
S3D_HD float s3d_expf_in_range(float x, const uint64_t *tab) { return s3d_expf_core(x, S3D_EXPF_FMA, tab); }

/* 3x3 symmetric eigen-decomposition, upper triangle of row-major A is read.
 * L ascending; eigenvector j is column j of row-major Q. */
S3D_HD void s3d_eigen3(const double *A, double *Q, double *L)
{
    double a00 = A[0], a01 = A[1], a02 = A[2], a11 = A[4], a12 = A[5], a22 = A[8];
    double v[3][3] = { { 1, 0, 0 }, { 0, 1, 0 }, { 0, 0, 1 } };
    double d[3];
    int ord[3] = { 0, 1, 2 };
    int sweep, i, j;
    for (sweep = 0; sweep < 64; sweep++) {
        int pq;
        if (fabs(a01) + fabs(a02) + fabs(a12) == 0.0)
            break;
        for (pq = 0; pq < 3; pq++) {
            /* rotation in plane (p,q): (0,1), (0,2), (1,2); k is the third index */
            const int p = pq == 2 ? 1 : 0, q = pq == 0 ? 1 : 2, k = 3 - p - q;
            double app, aqq, apq, akp, akq, theta, t, c, s;
            apq = pq == 0 ? a01 : pq == 1 ? a02 : a12;
            if (apq == 0.0)
                continue;
            app = p == 0 ? a00 : a11;
            aqq = q == 1 ? a11 : a22;
            theta = (aqq - app) / (2.0 * apq);
            t = (theta >= 0 ? 1.0 : -1.0) / (fabs(theta) + sqrt(theta * theta + 1.0));
            if (!(fabs(theta) <= 1.7976931348623157e308))
                t = 0.0;
            c = 1.0 / sqrt(t * t + 1.0);
            s = t * c;
            /* off-diagonal entries coupling the third index k */
            akp = (k == 0) ? (p == 1 ? a01 : a02) : (k == 1) ? (p == 0 ? a01 : a12)
                                                             : (p == 0 ? a02 : a12);
            akq = (k == 0) ? (q == 1 ? a01 : a02) : (k == 1) ? (q == 0 ? a01 : a12)
                                                             : (q == 0 ? a02 : a12);
            app = app - t * apq;
            aqq = aqq + t * apq;
            {
                const double nkp = c * akp - s * akq;
                const double nkq = s * akp + c * akq;
                if (pq == 0) { a00 = app; a11 = aqq; a01 = 0.0; a02 = nkp; a12 = nkq; }
                else if (pq == 1) { a00 = app; a22 = aqq; a02 = 0.0; a01 = nkp; a12 = nkq; }
                else { a11 = app; a22 = aqq; a12 = 0.0; a01 = nkp; a02 = nkq; }
            }
            for (i = 0; i < 3; i++) {
                const double vp = v[i][p], vq = v[i][q];
                v[i][p] = c * vp - s * vq;
                v[i][q] = s * vp + c * vq;
            }
        }
    }
    d[0] = a00; d[1] = a11; d[2] = a22;
    for (i = 0; i < 2; i++)
        for (j = 0; j < 2 - i; j++)
            if (d[ord[j]] > d[ord[j + 1]]) {
                const int tmp = ord[j];
                ord[j] = ord[j + 1];
                ord[j + 1] = tmp;
            }
    for (j = 0; j < 3; j++) {
        L[j] = d[ord[j]];
        for (i = 0; i < 3; i++)
            Q[3 * i + j] = v[i][ord[j]];
    }
}

#endif
