/* sift3d_oracle.c -- TEST INFRASTRUCTURE ONLY (see sift3d_oracle.h).
 *
 * A from-scratch CPU restatement of the numerical behaviour of the reference's
 * detect+describe path (fatimp/SIFT3D v2.0).  Citations are file:line relative
 * to /root/reference/sift3d/.  The structure is this repository's own (direct
 * strided 1-D passes instead of permute copies, flat level tables, no Mat_rm,
 * an in-line 3x3 Jacobi solver instead of LAPACK); the ARITHMETIC -- operation
 * order, intermediate types, comparison strictness, quirks Q1..Q9 of SURVEY.md
 * appendix A.6 -- is the reference's, so that outputs are bit-identical for
 * the float32 pyramid / candidate list / keypoint list and agree to rounding
 * noise (<= 1e-6 rel.) for R and descriptors (the only non-literal pieces are
 * the eigen-solver and the host libm).
 *
 * Build WITHOUT -march/-mfma and with -ffp-contract=off (oracle/Makefile): the
 * reference's release build has no FMA contraction.
 */
#include <float.h>
#include <math.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <time.h>

#include "sift3d_oracle.h"

#define ORC_NFACES 20
#define ORC_NVERT 12
#define ORC_NCELL 4                       /* NHIST_PER_DIM, imtypes_private.h:41 */
#define ORC_DESC_NUMEL (ORC_NCELL * ORC_NCELL * ORC_NCELL * ORC_NVERT)
#define ORC_SLAB 500                      /* SIFT3D_SLAB_LEN, immacros.h:199-201 */

/* sift.c:31-48 */
static const double k_peak_thresh_default = 0.1;
static const int k_num_kp_levels_default = 3;
static const double k_corner_thresh_default = 0.4;
static const double k_sigma_n_default = 1.15;
static const double k_sigma0_default = 1.6;
static const double k_max_eig_ratio = 0.90;
static const double k_ori_grad_thresh = 1E-10;
static const double k_bary_eps = FLT_EPSILON * 1E1;
static const double k_ori_sig_fctr = 1.5;
static const double k_ori_rad_fctr = 3.0;
static const double k_desc_sig_fctr = 7.071067812;
static const double k_desc_rad_fctr = 2.0;
static const double k_trunc_thresh = 0.2f * 128.0f / ORC_DESC_NUMEL;
static const double k_golden = 1.6180339887;

typedef struct {
    float *data;
    int nx, ny, nz;
    double ux, uy, uz;
    double s;
    /* Z-slab views (orc_orient_slab / orc_describe_slab): data holds planes
     * [z_off, z_off + nz) of a level with nz_glob planes.  0 / 0 = whole level. */
    int z_off, nz_glob;
} orc_level_t;

typedef struct {
    double sigma;
    int width;
    float *taps;
} orc_filter_t;

typedef struct {
    float v[3][3];
    int idx[3];
} orc_tri;

struct orc_ctx {
    /* parameters */
    double peak_thresh, corner_thresh, sigma_n, sigma0;
    int num_kp_levels;
    int fir_mode;
    int cuboid;             /* CUBOID_EXTREMA switch */
    /* scaled input copy (sift.c:645-649) */
    orc_level_t im;
    int have_im;
    /* pyramids: gpyr has num_kp_levels+3 levels s=-1.., dog num_kp_levels+2 */
    int num_octaves;
    int ngl, ndl;
    orc_level_t *gpyr, *dog;
    float *dogmax;                         /* [num_octaves][ndl] */
    /* filters: [0] first blur, [1..ngl-1] octave filters (make_gss) */
    orc_filter_t *filt;
    int nfilt;
    orc_tri mesh[ORC_NFACES];
    /* results */
    orc_candidate *cand;
    int ncand, cand_cap;
    orc_keypoint *kp;
    int nkp, kp_cap;
    orc_descriptor *desc;
    int ndesc;
    double t[6];
};

static double now_s(void)
{
    struct timespec t;
    clock_gettime(CLOCK_MONOTONIC, &t);
    return t.tv_sec + 1e-9 * t.tv_nsec;
}

/* ------------------------------------------------------------------------ */
/* Gaussian taps -- init_Gauss_filter, imutil.c:1267-1319                    */
/* ------------------------------------------------------------------------ */
static int gauss_half_width(double sigma)
{
    /* imutil.c:1275-1277, SIFT3D_GAUSS_WIDTH_FCTR = 3.0 */
    if (sigma > 0) {
        const int hw = (int)ceil(sigma * 3.0);
        return hw > 1 ? hw : 1;
    }
    return 1;
}

int orc_gauss_taps(double sigma, float *taps, int max_taps)
{
    const int hw = gauss_half_width(sigma);
    const int width = 2 * hw + 1;
    float acc = 0;
    int i;
    if (width > max_taps)
        return width;
    for (i = 0; i < width; i++) {
        double x = (double)i - hw;         /* imutil.c:1288 */
        x /= sigma + DBL_EPSILON;          /* imutil.c:1291 (Q5) */
        taps[i] = (float)exp(-0.5 * x * x);
        acc += taps[i];                    /* float running sum (Q5) */
    }
    for (i = 0; i < width; i++)
        taps[i] /= acc;
    return width;
}

/* ------------------------------------------------------------------------ */
/* 1-D interpolating FIR -- convolve_sep_gen, imutil.c:742-861               */
/* ------------------------------------------------------------------------ */

/* One output sample, literal.  line = pointer to GLOBAL index 0 of this 1-D
 * line (may be outside the local allocation; only [lo_ok, hi_ok] is read),
 * stride in floats.  Returns 0, or 1 if a needed sample was unavailable. */
static inline int fir_sample_literal(const float *line, ptrdiff_t stride,
                                     int g, int n_glob, const float *taps,
                                     int hw, float uf, int uhw, int lo_ok,
                                     int hi_ok, float *out)
{
    const int dim_end = n_glob - 1;                       /* imutil.c:753 */
    const int interior = g >= uhw && g <= n_glob - 2 - uhw; /* :762-763,829 */
    float acc = 0.0f;                                     /* im_zero, :777 */
    float coord = (float)g;                               /* :803 / :835   */
    int bad = 0;
    int d;
    for (d = -hw; d <= hw; d++) {
        const float tap = taps[d + hw];
        const float step = d * uf;                        /* :808 / :837   */
        float c, frac, a, b;
        int lo, hi;
        if (interior) {
            coord -= step;                                /* :811 */
            c = coord;
        } else {
            c = (float)g - step;                          /* :835,840 */
            if ((int)c < 0)                               /* :843 */
                c = -c;
            else if ((int)c >= dim_end)                   /* :846 */
                c = 2.0f * dim_end - c - 0.1f;            /* :847-848 (Q4) */
        }
        lo = (int)c;                                      /* :783 (trunc)  */
        hi = lo + 1;                                      /* :787 */
        frac = c - (float)lo;                             /* :788 */
        /* The reference reads src[hi] even when its weight is exactly 0 and
         * hi is one past the row (hw >= n corner case); give that
         * term weight 0 without the read.  Degenerate inputs whose mirrored
         * index leaves the row altogether (a dimension of exactly 8 with the
         * 17-tap filter) are undefined behaviour in the reference; we clamp. */
        if (lo < lo_ok || lo > hi_ok) {
            /* inside the global axis but outside this slab: halo too thin */
            if (lo >= 0 && lo < n_glob)
                bad = 1;
            lo = lo < lo_ok ? lo_ok : hi_ok;
        }
        a = line[(ptrdiff_t)lo * stride];
        if (frac == 0.0f && (hi < lo_ok || hi > hi_ok)) {
            b = 0.0f;                                     /* weight-0 term */
        } else {
            if (hi < lo_ok || hi > hi_ok) {
                if (hi >= 0 && hi < n_glob)
                    bad = 1;
                hi = hi < lo_ok ? lo_ok : hi_ok;
            }
            b = line[(ptrdiff_t)hi * stride];
        }
        acc += tap * ((1.0f - frac) * a + frac * b);      /* :791-795 */
        if (interior)
            coord += step;                                /* :817 */
    }
    *out = acc;
    return bad;
}

static int is_dyadic(float uf, int *shift)
{
    int e;
    const float m = frexpf(uf, &e);
    if (m != 0.5f || e > 1)
        return 0;
    *shift = 1 - e; /* uf = 2^-shift */
    return 1;
}

int orc_fir_axis(const float *src, float *dst, int nx, int ny, int nz, int axis,
                 const float *taps, int width, float uf, int n_glob, int off,
                 int out_lo, int out_hi, int mode)
{
    const int hw = width / 2;                              /* imutil.c:748 */
    const int uhw = (int)ceilf(hw * uf);                   /* :756-757 */
    const int dims[3] = { nx, ny, nz };
    const ptrdiff_t strides[3] = { 1, nx, (ptrdiff_t)nx * ny };
    const int n_loc = dims[axis];
    const ptrdiff_t st = strides[axis];
    const int lo_ok = off, hi_ok = off + n_loc - 1;
    int bad = 0;
    int shift = 0;
    /* restructured path: per-tap constant (offset, frac) -- valid when
     * (float)g - d*uf is exact, i.e. uf = 2^-k and g < 2^(23-k) */
    const int fast = mode == 1 && is_dyadic(uf, &shift) && shift <= 12 &&
                     n_glob < (1 << (23 - shift)) && hw < 1024 && axis != 0;
    int *toff = NULL;
    float *tw0 = NULL, *tw1 = NULL;

    if (fast) {
        int d;
        toff = (int *)malloc(sizeof(int) * width);
        tw0 = (float *)malloc(sizeof(float) * width);
        tw1 = (float *)malloc(sizeof(float) * width);
        for (d = -hw; d <= hw; d++) {
            const int g0 = 1 << 10; /* reference index: g0 +- hw*uf exact in float */
            const float c = (float)g0 - d * uf;
            const int lo = (int)c;
            const float frac = c - (float)lo;
            toff[d + hw] = lo - g0;
            tw0[d + hw] = 1.0f - frac;
            tw1[d + hw] = frac;
        }
    }

    if (axis == 0) {
        /* lines along x; parallel over (y,z) rows */
        const long nrows = (long)ny * nz;
        long r;
#pragma omp parallel for schedule(static) reduction(| : bad)
        for (r = 0; r < nrows; r++) {
            const float *line = src + r * nx - off;
            float *o = dst + r * nx;
            int p;
            for (p = out_lo; p < out_hi; p++)
                bad |= fir_sample_literal(line, 1, p + off, n_glob, taps, hw, uf,
                                          uhw, lo_ok, hi_ok, &o[p]);
        }
    } else {
        /* sweep axis y or z; inner loop over the contiguous x (and y) run */
        const int outer_n = axis == 1 ? nz : 1;
        const ptrdiff_t outer_st = axis == 1 ? strides[2] : 0;
        const long run = axis == 1 ? nx : (long)nx * ny;
        const long jobs = (long)outer_n * (out_hi - out_lo);
        long job;
#pragma omp parallel for schedule(static) reduction(| : bad)
        for (job = 0; job < jobs; job++) {
            const int ou = (int)(job / (out_hi - out_lo));
            const int p = out_lo + (int)(job % (out_hi - out_lo));
            const int g = p + off;
            const float *base = src + ou * outer_st;          /* local idx 0 */
            float *o = dst + ou * outer_st + (ptrdiff_t)p * st;
            const int interior = g >= uhw && g <= n_glob - 2 - uhw;
            long i;
            if (fast && interior) {
                int t;
                for (i = 0; i < run; i++)
                    o[i] = 0.0f;
                for (t = 0; t < width; t++) {
                    const float tap = taps[t];
                    const float w0 = tw0[t], w1 = tw1[t];
                    const int lo = g + toff[t] - off;     /* local index */
                    const float *a, *b;
                    if (lo < 0 || lo + 1 >= n_loc) {
                        /* hi may legitimately be unused when w1 == 0 */
                        if (lo < 0 || lo >= n_loc || w1 != 0.0f) { bad |= 1; continue; }
                        a = base + (ptrdiff_t)lo * st;
                        for (i = 0; i < run; i++)
                            o[i] += tap * (w0 * a[i] + w1 * 0.0f);
                        continue;
                    }
                    a = base + (ptrdiff_t)lo * st;
                    b = a + st;
                    for (i = 0; i < run; i++)
                        o[i] += tap * (w0 * a[i] + w1 * b[i]);
                }
            } else {
                for (i = 0; i < run; i++)
                    bad |= fir_sample_literal(base + i - (ptrdiff_t)off * st, st, g,
                                              n_glob, taps, hw, uf, uhw, lo_ok,
                                              hi_ok, &o[i]);
            }
        }
    }
    free(toff);
    free(tw0);
    free(tw1);
    return bad ? ORC_FAILURE : ORC_SUCCESS;
}

/* apply_Sep_FIR_filter, imutil.c:1127-1206 */
int orc_blur(const float *src, float *dst, int nx, int ny, int nz, double ux,
             double uy, double uz, const float *taps, int width, double unit,
             int mode)
{
    const size_t n = (size_t)nx * ny * nz;
    const double units[3] = { ux, uy, uz };
    const int dims[3] = { nx, ny, nz };
    float *tmp = (float *)malloc(n * sizeof(float));
    float *tmp2 = (float *)malloc(n * sizeof(float));
    const float *in = src;
    float *outs[3];
    int ax, ret = ORC_SUCCESS;
    if (!tmp || !tmp2) {
        free(tmp);
        free(tmp2);
        return ORC_FAILURE;
    }
    outs[0] = tmp;
    outs[1] = tmp2;
    outs[2] = dst;
    for (ax = 0; ax < 3; ax++) {
        /* imutil.c:1168-1169 (unit -1 = image units), :754-755 */
        const double unit_arg = unit == -1.0 ? units[ax] : unit;
        const float uf = unit_arg / units[ax];
        if (orc_fir_axis(in, outs[ax], nx, ny, nz, ax, taps, width, uf, dims[ax],
                         0, 0, dims[ax], mode))
            ret = ORC_FAILURE;
        in = outs[ax];
    }
    free(tmp);
    free(tmp2);
    return ret;
}

/* im_downsample_2x, imutil.c:591-617: dst(x,y,z) = src(2x,2y,2z) */
void orc_downsample(const float *src, int nx, int ny, int nz, float *dst)
{
    const int mx = nx / 2, my = ny / 2, mz = nz / 2; /* floor(n/2), :597-599 */
    int x, y, z;
    for (z = 0; z < mz; z++)
        for (y = 0; y < my; y++)
            for (x = 0; x < mx; x++)
                dst[x + (size_t)mx * (y + (size_t)my * z)] =
                    src[2 * x + (size_t)nx * (2 * y + (size_t)ny * (2 * z))];
}

/* ------------------------------------------------------------------------ */
/* 3x3 symmetric eigen-solver (stands in for LAPACK dsyevd, imutil.c:984)    */
/* ------------------------------------------------------------------------ */
void orc_eigen3(const double *A9, double *Q9, double *L3)
{
    double a[3][3], v[3][3];
    int i, j, sweep, order[3];
    for (i = 0; i < 3; i++)
        for (j = 0; j < 3; j++) {
            /* dsyevd('U') reads the upper triangle only (imutil.c:994) */
            a[i][j] = i <= j ? A9[3 * i + j] : A9[3 * j + i];
            v[i][j] = i == j;
        }
    for (sweep = 0; sweep < 64; sweep++) {
        const double offd = fabs(a[0][1]) + fabs(a[0][2]) + fabs(a[1][2]);
        int p, q;
        if (offd == 0.0)
            break;
        for (p = 0; p < 2; p++)
            for (q = p + 1; q < 3; q++) {
                double theta, t, c, s, app, aqq, apq;
                int k;
                apq = a[p][q];
                if (apq == 0.0)
                    continue;
                app = a[p][p];
                aqq = a[q][q];
                theta = (aqq - app) / (2.0 * apq);
                t = (theta >= 0 ? 1.0 : -1.0) /
                    (fabs(theta) + sqrt(theta * theta + 1.0));
                if (!isfinite(theta))
                    t = 0.0;
                c = 1.0 / sqrt(t * t + 1.0);
                s = t * c;
                a[p][p] = app - t * apq;
                a[q][q] = aqq + t * apq;
                a[p][q] = a[q][p] = 0.0;
                for (k = 0; k < 3; k++) {
                    if (k != p && k != q) {
                        const double akp = a[k][p], akq = a[k][q];
                        a[k][p] = a[p][k] = c * akp - s * akq;
                        a[k][q] = a[q][k] = s * akp + c * akq;
                    }
                }
                for (k = 0; k < 3; k++) {
                    const double vkp = v[k][p], vkq = v[k][q];
                    v[k][p] = c * vkp - s * vkq;
                    v[k][q] = s * vkp + c * vkq;
                }
            }
    }
    /* ascending eigenvalues; eigenvectors are the matching columns */
    order[0] = 0; order[1] = 1; order[2] = 2;
    for (i = 0; i < 2; i++)
        for (j = 0; j < 2 - i; j++)
            if (a[order[j]][order[j]] > a[order[j + 1]][order[j + 1]]) {
                const int t = order[j];
                order[j] = order[j + 1];
                order[j + 1] = t;
            }
    for (j = 0; j < 3; j++) {
        L3[j] = a[order[j]][order[j]];
        for (i = 0; i < 3; i++)
            Q9[3 * i + j] = v[i][order[j]];
    }
}

/* ------------------------------------------------------------------------ */
/* Icosahedron -- init_geometry, sift.c:148-259                              */
/* ------------------------------------------------------------------------ */
static void build_mesh(orc_tri *mesh)
{
    const float g = k_golden;
    const float vert[ORC_NVERT][3] = {
        { 0, 1, g }, { 0, -1, g }, { 0, 1, -g }, { 0, -1, -g },
        { 1, g, 0 }, { -1, g, 0 }, { 1, -g, 0 }, { -1, -g, 0 },
        { g, 0, 1 }, { -g, 0, 1 }, { g, 0, -1 }, { -g, 0, -1 } };
    static const int faces[ORC_NFACES][3] = {
        { 0, 1, 8 }, { 0, 8, 4 }, { 0, 4, 5 }, { 0, 5, 9 }, { 0, 9, 1 },
        { 1, 6, 8 }, { 8, 6, 10 }, { 8, 10, 4 }, { 4, 10, 2 }, { 4, 2, 5 },
        { 5, 2, 11 }, { 5, 11, 9 }, { 9, 11, 7 }, { 9, 7, 1 }, { 1, 7, 6 },
        { 3, 6, 7 }, { 3, 7, 11 }, { 3, 11, 2 }, { 3, 2, 10 }, { 3, 10, 6 } };
    int i, j, k;
    for (i = 0; i < ORC_NFACES; i++) {
        orc_tri *t = mesh + i;
        float e21[3], e10[3], n[3];
        for (j = 0; j < 3; j++) {
            float mag;
            t->idx[j] = faces[i][j];
            for (k = 0; k < 3; k++)
                t->v[j][k] = vert[faces[i][j]][k];
            mag = sqrtf(t->v[j][0] * t->v[j][0] + t->v[j][1] * t->v[j][1] +
                        t->v[j][2] * t->v[j][2]);           /* sift.c:226 */
            /* SIFT3D_CVEC_SCALE(v, 1.0f / mag) expands textually to
             * x = x * 1.0f / mag, i.e. a division (sift.c:228, immacros.h:285) */
            for (k = 0; k < 3; k++)
                t->v[j][k] = t->v[j][k] * 1.0f / mag;
        }
        for (k = 0; k < 3; k++) {
            e21[k] = t->v[2][k] - t->v[1][k];               /* sift.c:232 */
            e10[k] = t->v[1][k] - t->v[0][k];               /* sift.c:233 */
        }
        n[0] = e21[1] * e10[2] - e21[2] * e10[1];
        n[1] = e21[2] * e10[0] - e21[0] * e10[2];
        n[2] = e21[0] * e10[1] - e21[1] * e10[0];
        if (n[0] * t->v[0][0] + n[1] * t->v[0][1] + n[2] * t->v[0][2] < 0) {
            /* swap the first two VERTICES but not idx[] (Q1, sift.c:237-241) */
            for (k = 0; k < 3; k++) {
                const float tmp = t->v[0][k];
                t->v[0][k] = t->v[1][k];
                t->v[1][k] = tmp;
            }
        }
    }
}

/* cart2bary, sift.c:268-327 (Moller-Trumbore).  Returns 0 on success. */
static int ray_bary(const float *c, const orc_tri *t, float *bary, float *k)
{
    float e1[3], e2[3], p[3], q[3], tv[3], det, det_inv;
    int i;
    for (i = 0; i < 3; i++) {
        e1[i] = t->v[1][i] - t->v[0][i];
        e2[i] = t->v[2][i] - t->v[0][i];
    }
    p[0] = c[1] * e2[2] - c[2] * e2[1];
    p[1] = c[2] * e2[0] - c[0] * e2[2];
    p[2] = c[0] * e2[1] - c[1] * e2[0];
    det = e1[0] * p[0] + e1[1] * p[1] + e1[2] * p[2];
    if (fabsf(det) < k_bary_eps)                            /* sift.c:282 */
        return 1;
    det_inv = 1.0f / det;
    for (i = 0; i < 3; i++)
        tv[i] = t->v[0][i] * -1.0f;                         /* sift.c:288-289 */
    q[0] = tv[1] * e1[2] - tv[2] * e1[1];
    q[1] = tv[2] * e1[0] - tv[0] * e1[2];
    q[2] = tv[0] * e1[1] - tv[1] * e1[0];
    bary[1] = det_inv * (tv[0] * p[0] + tv[1] * p[1] + tv[2] * p[2]);
    bary[2] = det_inv * (c[0] * q[0] + c[1] * q[1] + c[2] * q[2]);
    bary[0] = 1.0f - bary[1] - bary[2];
    *k = (e2[0] * q[0] + e2[1] * q[1] + e2[2] * q[2]) * det_inv;
    return 0;
}

/* icos_hist_bin, sift.c:1254-1291.  Returns the face or -1. */
static int icos_bin(const orc_tri *mesh, const float *g, float *bary)
{
    int i;
    if (g[0] * g[0] + g[1] * g[1] + g[2] * g[2] < k_bary_eps) /* :1264 */
        return -1;
    for (i = 0; i < ORC_NFACES; i++) {
        float k;
        if (ray_bary(g, mesh + i, bary, &k))
            continue;
        if (bary[0] < -k_bary_eps || bary[1] < -k_bary_eps ||
            bary[2] < -k_bary_eps || k < 0)                 /* :1277-1279 */
            continue;
        return i;
    }
    return -1;
}

/* ------------------------------------------------------------------------ */
/* context, parameters, pyramid geometry                                     */
/* ------------------------------------------------------------------------ */
static void free_levels(orc_level_t *lv, int n)
{
    int i;
    if (!lv)
        return;
    for (i = 0; i < n; i++)
        free(lv[i].data);
    free(lv);
}

static void free_filters(orc_ctx *c)
{
    int i;
    for (i = 0; i < c->nfilt; i++)
        free(c->filt[i].taps);
    free(c->filt);
    c->filt = NULL;
    c->nfilt = 0;
}

orc_ctx *orc_create(void)
{
    orc_ctx *c = (orc_ctx *)calloc(1, sizeof(*c));
    if (!c)
        return NULL;
    c->peak_thresh = k_peak_thresh_default;
    c->corner_thresh = k_corner_thresh_default;
    c->sigma_n = k_sigma_n_default;
    c->sigma0 = k_sigma0_default;
    c->num_kp_levels = k_num_kp_levels_default;
    c->fir_mode = 1;
    build_mesh(c->mesh);
    return c;
}

void orc_destroy(orc_ctx *c)
{
    if (!c)
        return;
    free(c->im.data);
    free_levels(c->gpyr, c->num_octaves * c->ngl);
    free_levels(c->dog, c->num_octaves * c->ndl);
    free(c->dogmax);
    free_filters(c);
    free(c->cand);
    free(c->kp);
    free(c->desc);
    free(c);
}

void orc_set_fir_mode(orc_ctx *c, int mode) { c->fir_mode = mode; }

/* CUBOID_EXTREMA (sift.c:24) as a switch: 0 = default build, 1 = 80-neighbour test */
void orc_set_cuboid_extrema(orc_ctx *c, int on) { c->cuboid = on ? 1 : 0; }

static double level_scale(const orc_ctx *c, int o, int s)
{
    /* set_scales_Pyramid, imutil.c:1578-1579 */
    return c->sigma0 * pow(2.0, o + (double)s / c->num_kp_levels);
}

/* make_gss, imutil.c:1360-1409 + init_Gauss_incremental_filter :1322-1343 */
static int build_filters(orc_ctx *c)
{
    const int nf = c->num_kp_levels + 3; /* 1 + (num_gpyr_levels - 1) */
    int i;
    free_filters(c);
    c->filt = (orc_filter_t *)calloc(nf, sizeof(orc_filter_t));
    c->nfilt = nf;
    for (i = 0; i < nf; i++) {
        const double s_cur = i == 0 ? c->sigma_n : level_scale(c, 0, i - 2);
        const double s_next = level_scale(c, 0, i - 1);
        double sigma;
        int w;
        if (s_cur > s_next) {
            fprintf(stderr, "orc: s_cur (%f) > s_next (%f)\n", s_cur, s_next);
            return ORC_FAILURE;
        }
        sigma = sqrt(s_next * s_next - s_cur * s_cur);
        w = 2 * gauss_half_width(sigma) + 1;
        c->filt[i].sigma = sigma;
        c->filt[i].width = w;
        c->filt[i].taps = (float *)malloc(sizeof(float) * w);
        orc_gauss_taps(sigma, c->filt[i].taps, w);
    }
    return ORC_SUCCESS;
}

/* resize_SIFT3D (sift.c:427-475) + resize_Pyramid (imutil.c:1464-1554) */
static int resize_pyramids(orc_ctx *c)
{
    const int ngl = c->num_kp_levels + 3, ndl = c->num_kp_levels + 2;
    int mn, last_octave, o, s, dims[3];
    double units[3];

    free_levels(c->gpyr, c->num_octaves * c->ngl);
    free_levels(c->dog, c->num_octaves * c->ndl);
    free(c->dogmax);
    c->gpyr = c->dog = NULL;
    c->dogmax = NULL;
    c->num_octaves = 0;
    c->ngl = ngl;
    c->ndl = ndl;
    if (!c->have_im)
        return ORC_SUCCESS;

    mn = c->im.nx < c->im.ny ? c->im.nx : c->im.ny;
    mn = mn < c->im.nz ? mn : c->im.nz;
    last_octave = (int)log2((double)mn) - 3;                /* sift.c:442-444 */
    if (last_octave < 0) {
        fprintf(stderr, "orc: input image is too small: must have at least 8 "
                        "voxels in each dimension\n");
        return ORC_FAILURE;
    }
    /* sigma_n check of set_scales_Pyramid, imutil.c:1582-1588 */
    if (level_scale(c, 0, -1) < c->sigma_n) {
        fprintf(stderr, "orc: sigma_n too large for these settings\n");
        return ORC_FAILURE;
    }
    c->num_octaves = last_octave + 1;
    c->gpyr = (orc_level_t *)calloc((size_t)c->num_octaves * ngl, sizeof(orc_level_t));
    c->dog = (orc_level_t *)calloc((size_t)c->num_octaves * ndl, sizeof(orc_level_t));
    c->dogmax = (float *)calloc((size_t)c->num_octaves * ndl, sizeof(float));
    dims[0] = c->im.nx; dims[1] = c->im.ny; dims[2] = c->im.nz;
    units[0] = c->im.ux; units[1] = c->im.uy; units[2] = c->im.uz;
    for (o = 0; o < c->num_octaves; o++) {
        const size_t n = (size_t)dims[0] * dims[1] * dims[2];
        for (s = 0; s < ngl + ndl; s++) {
            orc_level_t *lv = s < ngl ? &c->gpyr[o * ngl + s]
                                      : &c->dog[o * ndl + (s - ngl)];
            const int lev = (s < ngl ? s : s - ngl) - 1;
            lv->nx = dims[0]; lv->ny = dims[1]; lv->nz = dims[2];
            lv->ux = units[0]; lv->uy = units[1]; lv->uz = units[2];
            lv->s = level_scale(c, o, lev);
            lv->data = (float *)malloc(n * sizeof(float));
            if (!lv->data)
                return ORC_FAILURE;
        }
        for (s = 0; s < 3; s++) {                           /* imutil.c:1545-1548 */
            dims[s] /= 2;
            units[s] *= 2;
        }
    }
    return build_filters(c);
}

int orc_set_peak_thresh(orc_ctx *c, double v)
{
    if (v <= 0.0 || v > 1) {                                /* sift.c:501 */
        fprintf(stderr, "orc: peak_thresh must be in the interval (0, 1]\n");
        return ORC_FAILURE;
    }
    c->peak_thresh = v;
    return ORC_SUCCESS;
}

int orc_set_corner_thresh(orc_ctx *c, double v)
{
    if (v < 0.0 || v > 1.0) {                               /* sift.c:515 */
        fprintf(stderr, "orc: corner_thresh must be in the interval [0, 1]\n");
        return ORC_FAILURE;
    }
    c->corner_thresh = v;
    return ORC_SUCCESS;
}

int orc_set_num_kp_levels(orc_ctx *c, unsigned v)
{
    c->num_kp_levels = (int)v;                              /* sift.c:527-533 */
    return resize_pyramids(c);
}

static int set_scales(orc_ctx *c, double sigma0, double sigma_n)
{
    /* set_scales_SIFT3D, sift.c:478-496 (check happens only with levels) */
    const double old0 = c->sigma0, oldn = c->sigma_n;
    int o, s;
    c->sigma0 = sigma0;
    c->sigma_n = sigma_n;
    if (!c->num_octaves)
        return ORC_SUCCESS;
    if (level_scale(c, 0, -1) < sigma_n) {
        c->sigma0 = old0;
        c->sigma_n = oldn;
        fprintf(stderr, "orc: sigma_n too large for these settings\n");
        return ORC_FAILURE;
    }
    for (o = 0; o < c->num_octaves; o++) {
        for (s = 0; s < c->ngl; s++)
            c->gpyr[o * c->ngl + s].s = level_scale(c, o, s - 1);
        for (s = 0; s < c->ndl; s++)
            c->dog[o * c->ndl + s].s = level_scale(c, o, s - 1);
    }
    return build_filters(c);
}

int orc_set_sigma_n(orc_ctx *c, double v)
{
    if (v < 0.0) {                                          /* sift.c:542 */
        fprintf(stderr, "orc: sigma_n must be nonnegative\n");
        return ORC_FAILURE;
    }
    return set_scales(c, c->sigma0, v);
}

int orc_set_sigma0(orc_ctx *c, double v)
{
    if (v < 0.0) {                                          /* sift.c:558 */
        fprintf(stderr, "orc: sigma0 must be nonnegative\n");
        return ORC_FAILURE;
    }
    return set_scales(c, v, c->sigma_n);
}

/* set_im_SIFT3D, sift.c:629-659: copy, scale by max|v|, resize on new dims */
int orc_set_volume(orc_ctx *c, const float *vol, int nx, int ny, int nz,
                   double ux, double uy, double uz)
{
    const size_t n = (size_t)nx * ny * nz;
    const int dims_changed = !c->have_im || c->im.nx != nx || c->im.ny != ny ||
                             c->im.nz != nz;
    const double t0 = now_s();
    float mx = 0.0f;
    size_t i;
    if (nx < 1 || ny < 1 || nz < 1)
        return ORC_FAILURE;
    if (dims_changed) {
        free(c->im.data);
        c->im.data = (float *)malloc(n * sizeof(float));
        if (!c->im.data)
            return ORC_FAILURE;
    }
    c->im.nx = nx; c->im.ny = ny; c->im.nz = nz;
    c->im.ux = ux; c->im.uy = uy; c->im.uz = uz;
    c->have_im = 1;
    for (i = 0; i < n; i++) {                               /* imutil.c:681-695 */
        const float a = fabsf(vol[i]);
        mx = mx > a ? mx : a;
    }
    if (mx == 0.0f)
        memcpy(c->im.data, vol, n * sizeof(float));         /* imutil.c:706-707 */
    else
        for (i = 0; i < n; i++)
            c->im.data[i] = vol[i] / mx;                    /* imutil.c:711 */
    c->t[0] = now_s() - t0;
    /* NB the reference keeps level units from the image that triggered the
     * last resize (resize_Pyramid copies units only then, sift.c:652-656) */
    if (dims_changed)
        return resize_pyramids(c);
    return ORC_SUCCESS;
}

static orc_level_t *G(const orc_ctx *c, int o, int s) { return &c->gpyr[o * c->ngl + s + 1]; }
static orc_level_t *D(const orc_ctx *c, int o, int s) { return &c->dog[o * c->ndl + s + 1]; }

/* build_gpyr (sift.c:662-711) + build_dog (sift.c:713-732) */
int orc_build_pyramids(orc_ctx *c)
{
    int o, s;
    double t0 = now_s();
    if (!c->num_octaves)
        return ORC_FAILURE;
    {
        orc_level_t *g = G(c, 0, -1);
        /* apply_Sep_FIR_filter copies the units of its source (imutil.c:1145) */
        if (orc_blur(c->im.data, g->data, g->nx, g->ny, g->nz, c->im.ux, c->im.uy,
                     c->im.uz, c->filt[0].taps, c->filt[0].width, 1.0, c->fir_mode))
            return ORC_FAILURE;
        g->ux = c->im.ux; g->uy = c->im.uy; g->uz = c->im.uz;
    }
    for (o = 0; o < c->num_octaves; o++) {
        for (s = 0; s <= c->ngl - 2; s++) {
            const orc_level_t *p = G(c, o, s - 1);
            orc_level_t *g = G(c, o, s);
            const orc_filter_t *f = &c->filt[s + 1];       /* gauss_octave[s], sift.c:689 */
            if (orc_blur(p->data, g->data, g->nx, g->ny, g->nz, p->ux, p->uy, p->uz,
                         f->taps, f->width, 1.0, c->fir_mode))
                return ORC_FAILURE;
            g->ux = p->ux; g->uy = p->uy; g->uz = p->uz;
        }
        if (o != c->num_octaves - 1) {
            /* downsample level max(s_end - 2, first_level), sift.c:696-704 */
            const int s_end = c->ngl - 2;
            const int ds = s_end - 2 > -1 ? s_end - 2 : -1;
            const orc_level_t *p = G(c, o, ds);
            orc_downsample(p->data, p->nx, p->ny, p->nz, G(c, o + 1, -1)->data);
        }
    }
    c->t[1] = now_s() - t0;
    t0 = now_s();
    for (o = 0; o < c->num_octaves; o++)
        for (s = -1; s <= c->ndl - 2; s++) {
            const orc_level_t *a = G(c, o, s), *b = G(c, o, s + 1);
            orc_level_t *d = D(c, o, s);
            const size_t n = (size_t)a->nx * a->ny * a->nz;
            size_t i;
            for (i = 0; i < n; i++)
                d->data[i] = a->data[i] - b->data[i];       /* imutil.c:734-737 */
            d->ux = a->ux; d->uy = a->uy; d->uz = a->uz;    /* im_copy_dims */
        }
    c->t[2] = now_s() - t0;
    return ORC_SUCCESS;
}

/* CMP_CUBE over prev / cur (self ignored) / next, sift.c:761-796 */
static int cuboid_extremum(const float *pv, const float *cv, const float *nv, size_t p, size_t ys,
                           size_t zs, float v)
{
    int gt = 1, lt = 1, dx, dy, dz;
    for (dz = -1; dz <= 1; dz++)
        for (dy = -1; dy <= 1; dy++)
            for (dx = -1; dx <= 1; dx++) {
                const size_t q = (size_t)((long)p + dx + (long)ys * dy + (long)zs * dz);
                gt = gt && v > pv[q] && v > nv[q];
                lt = lt && v < pv[q] && v < nv[q];
                if (dx || dy || dz) {
                    gt = gt && v > cv[q];
                    lt = lt && v < cv[q];
                }
            }
    return gt || lt;
}

/* detect_extrema, sift.c:735-871 (default build: 8 neighbours, D2; c->cuboid: the
 * CUBOID_EXTREMA build) */
int orc_find_extrema(orc_ctx *c)
{
    int o, s;
    const double t0 = now_s();
    c->ncand = 0;
    if (c->ndl < 3)
        return ORC_FAILURE;
    for (o = 0; o < c->num_octaves; o++)
        for (s = 0; s <= c->ndl - 3; s++) {
            const orc_level_t *cur = D(c, o, s);
            const float *pv = D(c, o, s - 1)->data, *cv = cur->data,
                        *nv = D(c, o, s + 1)->data;
            const int nx = cur->nx, ny = cur->ny, nz = cur->nz;
            const size_t n = (size_t)nx * ny * nz, ys = nx, zs = (size_t)nx * ny;
            float dogmax = 0.0f, thr;
            size_t i;
            int x, y, z;
            for (i = 0; i < n; i++) {                       /* sift.c:822-826 */
                const float a = fabsf(cv[i]);
                dogmax = dogmax > a ? dogmax : a;
            }
            c->dogmax[o * c->ndl + s + 1] = dogmax;
            thr = c->peak_thresh * dogmax;                  /* sift.c:829 */
            for (z = 1; z <= nz - 2; z++)
                for (y = 1; y <= ny - 2; y++)
                    for (x = 1; x <= nx - 2; x++) {
                        const size_t p = x + ys * y + zs * z;
                        const float v = cv[p];
                        if (!(v > thr || v < -thr))         /* sift.c:842 */
                            continue;
                        if (c->cuboid) {
                            if (!cuboid_extremum(pv, cv, nv, p, ys, zs, v))
                                continue;
                        } else
                        if (!((v > pv[p] && v > cv[p + 1] && v > cv[p - 1] &&
                               v > cv[p + ys] && v > cv[p - ys] && v > cv[p - zs] &&
                               v > cv[p + zs] && v > nv[p]) ||
                              (v < pv[p] && v < cv[p + 1] && v < cv[p - 1] &&
                               v < cv[p + ys] && v < cv[p - ys] && v < cv[p - zs] &&
                               v < cv[p + zs] && v < nv[p])))   /* sift.c:844-849 */
                            continue;
                        if (c->ncand == c->cand_cap) {
                            c->cand_cap += ORC_SLAB;
                            c->cand = (orc_candidate *)realloc(
                                c->cand, sizeof(orc_candidate) * c->cand_cap);
                        }
                        c->cand[c->ncand].o = o;
                        c->cand[c->ncand].s = s;
                        c->cand[c->ncand].x = x;
                        c->cand[c->ncand].y = y;
                        c->cand[c->ncand].z = z;
                        c->cand[c->ncand].sd = cur->s;      /* sift.c:860 */
                        c->cand[c->ncand].strength = fabsf(v);
                        c->ncand++;
                    }
        }
    c->t[3] = now_s() - t0;
    return ORC_SUCCESS;
}

/* IM_LOOP_SPHERE_START bounds (sift.c:86-99).  `rad_is_double` selects the C
 * promotion of the macro's expressions: win_radius is double in
 * assign_eig_ori (sift.c:936) and float in extract_descrip (sift.c:1454). */
static void sphere_bounds(float c, double rad_d, float rad_f, int rad_is_double,
                          float uf, int n, int *start, int *end)
{
    float lo, hi;
    if (rad_is_double) {
        lo = floorf(c - rad_d / uf);
        hi = ceilf(c + rad_d / uf);
    } else {
        lo = floorf(c - rad_f / uf);
        hi = ceilf(c + rad_f / uf);
    }
    *start = lo > 1 ? lo : 1;                               /* SIFT3D_MAX(.., 1) */
    *end = hi < n - 2 ? hi : n - 2;                         /* SIFT3D_MIN(.., n-2) */
}

/* IM_GET_GRAD_ISO, sift.c:140-145 + immacros.h:105-111 */
static inline void grad_iso(const orc_level_t *im, int x, int y, int z, float *g)
{
    const size_t ys = im->nx, zs = (size_t)im->nx * im->ny;
    const float *p = im->data + x + ys * y + zs * (size_t)(z - im->z_off);
    g[0] = 0.5f * (p[1] - p[-1]);
    g[1] = 0.5f * (p[ys] - p[-(ptrdiff_t)ys]);
    g[2] = 0.5f * (p[zs] - p[-(ptrdiff_t)zs]);
    g[0] *= 1.0f / (float)im->ux;
    g[1] *= 1.0f / (float)im->uy;
    g[2] *= 1.0f / (float)im->uz;
}

/* assign_eig_ori + assign_orientation_thresh, sift.c:926-1102.
 * Returns 0 keep, 1 reject. */
static int orient_one(const orc_ctx *c, const orc_level_t *im, const float *ctr,
                      double sigma, float *R)
{
    const double win_radius = sigma * k_ori_rad_fctr;       /* sift.c:936 */
    const float uxf = (float)im->ux, uyf = (float)im->uy, uzf = (float)im->uz;
    double A[9] = { 0 }, Q[9], L[3], corner = DBL_MAX;
    float win[3] = { 0.0f, 0.0f, 0.0f }, v[2][3], vr[3];
    int xs, xe, ys, ye, zs, ze, x, y, z, i;

    if (sigma < 0)
        return 1;
    sphere_bounds(ctr[0], win_radius, 0, 1, uxf, im->nx, &xs, &xe);
    sphere_bounds(ctr[1], win_radius, 0, 1, uyf, im->ny, &ys, &ye);
    sphere_bounds(ctr[2], win_radius, 0, 1, uzf, im->nz_glob ? im->nz_glob : im->nz, &zs, &ze);
    for (z = zs; z <= ze; z++)
        for (y = ys; y <= ye; y++)
            for (x = xs; x <= xe; x++) {
                float disp[3], sq, w, g[3];
                disp[0] = ((float)x - ctr[0]) * uxf;        /* sift.c:102-104 */
                disp[1] = ((float)y - ctr[1]) * uyf;
                disp[2] = ((float)z - ctr[2]) * uzf;
                sq = disp[0] * disp[0] + disp[1] * disp[1] + disp[2] * disp[2];
                if (sq > win_radius * win_radius)           /* double compare */
                    continue;
                w = expf(-0.5 * sq / (sigma * sigma));      /* sift.c:972 */
                grad_iso(im, x, y, z, g);
                A[0] += (double)g[0] * g[0] * w;            /* sift.c:978-983 */
                A[1] += (double)g[0] * g[1] * w;
                A[2] += (double)g[0] * g[2] * w;
                A[4] += (double)g[1] * g[1] * w;
                A[5] += (double)g[1] * g[2] * w;
                A[8] += (double)g[2] * g[2] * w;
                g[0] = g[0] * w;                            /* sift.c:986-987 */
                g[1] = g[1] * w;
                g[2] = g[2] * w;
                win[0] = win[0] + g[0];
                win[1] = win[1] + g[1];
                win[2] = win[2] + g[2];
            }
    A[3] = A[1]; A[6] = A[2]; A[7] = A[5];
    if (win[0] * win[0] + win[1] * win[1] + win[2] * win[2] <
        (float)k_ori_grad_thresh)                           /* sift.c:997 */
        return 1;
    orc_eigen3(A, Q, L);
    for (i = 0; i < 2; i++)
        if (fabs(L[i] / L[i + 1]) > k_max_eig_ratio)        /* sift.c:1011-1015 */
            return 1;
    for (i = 0; i < 2; i++) {
        const int e = 2 - i;                                /* descending order */
        double d, cos_ang, ac;
        float sgn;
        vr[0] = (float)Q[0 * 3 + e];
        vr[1] = (float)Q[1 * 3 + e];
        vr[2] = (float)Q[2 * 3 + e];
        d = win[0] * vr[0] + win[1] * vr[1] + win[2] * vr[2]; /* float dot */
        cos_ang = d / (sqrtf(vr[0] * vr[0] + vr[1] * vr[1] + vr[2] * vr[2]) *
                       sqrtf(win[0] * win[0] + win[1] * win[1] + win[2] * win[2]));
        ac = fabs(cos_ang);
        corner = corner < ac ? corner : ac;                 /* sift.c:1036 */
        sgn = d > 0.0 ? 1.0f : -1.0f;
        vr[0] = vr[0] * sgn; vr[1] = vr[1] * sgn; vr[2] = vr[2] * sgn;
        R[0 * 3 + i] = vr[0]; R[1 * 3 + i] = vr[1]; R[2 * 3 + i] = vr[2];
        v[i][0] = vr[0]; v[i][1] = vr[1]; v[i][2] = vr[2];
    }
    R[0 * 3 + 2] = v[0][1] * v[1][2] - v[0][2] * v[1][1];   /* sift.c:1054-1059 */
    R[1 * 3 + 2] = v[0][2] * v[1][0] - v[0][0] * v[1][2];
    R[2 * 3 + 2] = v[0][0] * v[1][1] - v[0][1] * v[1][0];
    return corner < c->corner_thresh ? 1 : 0;               /* sift.c:1100-1101 */
}

/* assign_orientations, sift.c:1109-1167 */
int orc_assign_orientations(orc_ctx *c)
{
    const double t0 = now_s();
    unsigned char *keep;
    float *Rs;
    int i, j;
    if (c->ncand > c->kp_cap) {
        c->kp_cap = ((c->ncand + ORC_SLAB - 1) / ORC_SLAB) * ORC_SLAB;
        c->kp = (orc_keypoint *)realloc(c->kp, sizeof(orc_keypoint) * c->kp_cap);
    }
    keep = (unsigned char *)malloc(c->ncand + 1);
    Rs = (float *)malloc(sizeof(float) * 9 * (c->ncand + 1));
#pragma omp parallel for schedule(dynamic, 8)
    for (i = 0; i < c->ncand; i++) {
        const orc_candidate *k = c->cand + i;
        const float ctr[3] = { (float)(double)k->x, (float)(double)k->y,
                               (float)(double)k->z };       /* sift.c:1124 */
        const double sigma = k_ori_sig_fctr * k->sd;        /* sift.c:1125 */
        keep[i] = !orient_one(c, G(c, k->o, k->s), ctr, sigma, Rs + 9 * i);
    }
    /* in-place compaction with copy_Keypoint, which does NOT copy `strength`
     * (sift.c:372-384, 1148-1162): slot j keeps candidate j's strength (Q2) */
    for (i = 0, j = 0; i < c->ncand; i++) {
        const orc_candidate *k = c->cand + i;
        orc_keypoint *q;
        if (!keep[i])
            continue;
        q = c->kp + j;
        memcpy(q->R, Rs + 9 * i, sizeof(float) * 9);
        q->xd = k->x; q->yd = k->y; q->zd = k->z;
        q->sd = k->sd;
        q->o = k->o; q->s = k->s;
        q->strength = c->cand[j].strength;
        j++;
    }
    c->nkp = j;
    free(keep);
    free(Rs);
    c->t[4] = now_s() - t0;
    return ORC_SUCCESS;
}

int orc_detect(orc_ctx *c, const float *vol, int nx, int ny, int nz, double ux,
               double uy, double uz)
{
    if (orc_set_volume(c, vol, nx, ny, nz, ux, uy, uz) || orc_build_pyramids(c) ||
        orc_find_extrema(c) || orc_assign_orientations(c))
        return ORC_FAILURE;
    return ORC_SUCCESS;
}

/* normalize_desc, sift.c:1402-1429 */
static void normalize_hist(float *h)
{
    double norm = 0.0;
    float norm_inv;
    int i;
    for (i = 0; i < ORC_DESC_NUMEL; i++)
        norm += (double)h[i] * h[i];
    norm = sqrt(norm) + DBL_EPSILON;
    norm_inv = 1.0f / norm;
    for (i = 0; i < ORC_DESC_NUMEL; i++)
        h[i] *= norm_inv;
}

/* extract_descrip (sift.c:1442-1536) + SIFT3D_desc_acc_interp (sift.c:1295-1399) */
static void describe_one(const orc_ctx *c, const orc_level_t *im,
                         const orc_keypoint *key, orc_descriptor *desc)
{
    const float sigma = key->sd * k_desc_sig_fctr;          /* sift.c:1453 */
    const float win_radius = k_desc_rad_fctr * sigma;
    const float half_width = win_radius / sqrt(2);
    const float desc_width = 2.0f * half_width;
    const float hist_width = desc_width / ORC_NCELL;
    const float bin_fctr = 1.0f / hist_width;
    const double coord_factor = ldexp(1.0, key->o);
    const float uxf = (float)im->ux, uyf = (float)im->uy, uzf = (float)im->uz;
    const float ctr[3] = { (float)key->xd, (float)key->yd, (float)key->zd };
    const float *R = key->R; /* Rt[i][j] = R[j][i] */
    float *hist = desc->hist;
    int xs, xe, ys, ye, zs, ze, x, y, z, i;

    memset(hist, 0, sizeof(float) * ORC_DESC_NUMEL);
    sphere_bounds(ctr[0], 0, win_radius, 0, uxf, im->nx, &xs, &xe);
    sphere_bounds(ctr[1], 0, win_radius, 0, uyf, im->ny, &ys, &ye);
    sphere_bounds(ctr[2], 0, win_radius, 0, uzf, im->nz_glob ? im->nz_glob : im->nz, &zs, &ze);
    for (z = zs; z <= ze; z++)
        for (y = ys; y <= ye; y++)
            for (x = xs; x <= xe; x++) {
                float vim[3], vkp[3], vb[3], dvb[3], g[3], gr[3], bary[3];
                float sq, w, mag;
                int face, dx, dy, dz;
                vim[0] = ((float)x - ctr[0]) * uxf;
                vim[1] = ((float)y - ctr[1]) * uyf;
                vim[2] = ((float)z - ctr[2]) * uzf;
                sq = vim[0] * vim[0] + vim[1] * vim[1] + vim[2] * vim[2];
                if (sq > win_radius * win_radius)           /* float compare */
                    continue;
                /* vkp = Rt * vim, SIFT3D_MUL_MAT_RM_CVEC (immacros.h:328-340) */
                vkp[0] = R[0] * vim[0] + R[3] * vim[1] + R[6] * vim[2];
                vkp[1] = R[1] * vim[0] + R[4] * vim[1] + R[7] * vim[2];
                vkp[2] = R[2] * vim[0] + R[5] * vim[1] + R[8] * vim[2];
                vb[0] = (vkp[0] + half_width) * bin_fctr;   /* sift.c:1483-1485 */
                vb[1] = (vkp[1] + half_width) * bin_fctr;
                vb[2] = (vkp[2] + half_width) * bin_fctr;
                if (vb[0] < 0 || vb[1] < 0 || vb[2] < 0 || vb[0] >= (float)ORC_NCELL ||
                    vb[1] >= (float)ORC_NCELL || vb[2] >= (float)ORC_NCELL)
                    continue;
                grad_iso(im, x, y, z, g);
                w = expf(-0.5f * sq / (sigma * sigma));     /* sift.c:1498 */
                g[0] = g[0] * w; g[1] = g[1] * w; g[2] = g[2] * w;
                gr[0] = R[0] * g[0] + R[3] * g[1] + R[6] * g[2];
                gr[1] = R[1] * g[0] + R[4] * g[1] + R[7] * g[2];
                gr[2] = R[2] * g[0] + R[5] * g[1] + R[8] * g[2];
                /* SIFT3D_desc_acc_interp */
                dvb[0] = vb[0] - floorf(vb[0]);
                dvb[1] = vb[1] - floorf(vb[1]);
                dvb[2] = vb[2] - floorf(vb[2]);
                face = icos_bin(c->mesh, gr, bary);
                if (face < 0)
                    continue;
                mag = sqrtf(gr[0] * gr[0] + gr[1] * gr[1] + gr[2] * gr[2]);
                for (dx = 0; dx < 2; dx++)
                    for (dy = 0; dy < 2; dy++)
                        for (dz = 0; dz < 2; dz++) {
                            const int cx = (int)vb[0] + dx, cy = (int)vb[1] + dy,
                                      cz = (int)vb[2] + dz;
                            float wt, *h;
                            if (cx < 0 || cx >= ORC_NCELL || cy < 0 || cy >= ORC_NCELL ||
                                cz < 0 || cz >= ORC_NCELL)
                                continue;
                            h = hist + (cx + cy * ORC_NCELL + cz * ORC_NCELL * ORC_NCELL) *
                                           ORC_NVERT;
                            wt = ((dx == 0) ? (1.0f - dvb[0]) : dvb[0]) *
                                 ((dy == 0) ? (1.0f - dvb[1]) : dvb[1]) *
                                 ((dz == 0) ? (1.0f - dvb[2]) : dvb[2]);
                            /* bins are addressed through the UNSWAPPED idx[] (Q1) */
                            h[c->mesh[face].idx[0]] += mag * wt * bary[0];
                            h[c->mesh[face].idx[1]] += mag * wt * bary[1];
                            h[c->mesh[face].idx[2]] += mag * wt * bary[2];
                        }
            }
    normalize_hist(hist);
    for (i = 0; i < ORC_DESC_NUMEL; i++)                    /* sift.c:1517-1523 */
        hist[i] = hist[i] < (float)k_trunc_thresh ? hist[i] : (float)k_trunc_thresh;
    normalize_hist(hist);
    desc->xd = key->xd * coord_factor;
    desc->yd = key->yd * coord_factor;
    desc->zd = key->zd * coord_factor;
    desc->sd = key->sd;
}

/* verify_keys (sift.c:1171-1212) + do_extract_descriptors (sift.c:1561-1596) */
int orc_describe(orc_ctx *c)
{
    const double t0 = now_s();
    int i;
    if (c->nkp < 1) {
        fprintf(stderr, "orc: invalid number of keypoints: %d\n", c->nkp);
        return ORC_FAILURE;
    }
    for (i = 0; i < c->nkp; i++) {
        const orc_keypoint *k = c->kp + i;
        const double f = ldexp(1.0, k->o);
        if (k->xd < 0 || k->yd < 0 || k->zd < 0 || k->xd * f >= (double)c->im.nx ||
            k->yd * f >= (double)c->im.ny || k->zd * f >= (double)c->im.nz || k->sd <= 0)
            return ORC_FAILURE;
        if (k->o < 0 || k->o >= c->num_octaves || k->s < -1 || k->s > c->ngl - 2)
            return ORC_FAILURE;
    }
    if (!c->num_octaves)
        return ORC_FAILURE;
    c->desc = (orc_descriptor *)realloc(c->desc, sizeof(orc_descriptor) * c->nkp);
    c->ndesc = c->nkp;
#pragma omp parallel for schedule(dynamic, 1)
    for (i = 0; i < c->nkp; i++)
        describe_one(c, G(c, c->kp[i].o, c->kp[i].s), c->kp + i, c->desc + i);
    c->t[5] = now_s() - t0;
    return ORC_SUCCESS;
}

/* keypoint_strength_cmp, sift.c:1832-1837 (never returns 0, Q7) */
static int strength_cmp(const void *a, const void *b)
{
    return (((const orc_keypoint *)a)->strength < ((const orc_keypoint *)b)->strength)
               ? 1 : -1;
}

void orc_sort_by_strength(orc_ctx *c, int limit)
{
    qsort(c->kp, c->nkp, sizeof(orc_keypoint), strength_cmp);
    if (c->nkp > limit && limit != 0)
        c->nkp = limit;
}

int orc_num_octaves(const orc_ctx *c) { return c->num_octaves; }
int orc_num_candidates(const orc_ctx *c) { return c->ncand; }
int orc_num_keypoints(const orc_ctx *c) { return c->nkp; }
int orc_num_descriptors(const orc_ctx *c) { return c->ndesc; }
const orc_candidate *orc_candidates(const orc_ctx *c) { return c->cand; }
const orc_keypoint *orc_keypoints(const orc_ctx *c) { return c->kp; }
const orc_descriptor *orc_descriptors(const orc_ctx *c) { return c->desc; }
const double *orc_timings(const orc_ctx *c) { return c->t; }

int orc_set_keypoints(orc_ctx *c, const orc_keypoint *k, int n)
{
    if (n > c->kp_cap) {
        c->kp_cap = n;
        c->kp = (orc_keypoint *)realloc(c->kp, sizeof(orc_keypoint) * n);
    }
    memcpy(c->kp, k, sizeof(orc_keypoint) * n);
    c->nkp = n;
    return ORC_SUCCESS;
}

const float *orc_level(const orc_ctx *c, int which, int o, int s, int *dims,
                       double *units, double *scale)
{
    const orc_level_t *lv = which == 2 ? &c->im : which == 1 ? D(c, o, s) : G(c, o, s);
    dims[0] = lv->nx; dims[1] = lv->ny; dims[2] = lv->nz;
    units[0] = lv->ux; units[1] = lv->uy; units[2] = lv->uz;
    *scale = which == 2 ? -1.0 : lv->s;
    return lv->data;
}

float orc_dogmax(const orc_ctx *c, int o, int s) { return c->dogmax[o * c->ndl + s + 1]; }

int orc_filter(const orc_ctx *c, int idx, double *sigma, float *taps)
{
    const orc_filter_t *f;
    if (idx + 1 < 0 || idx + 1 >= c->nfilt)
        return -1;
    f = &c->filt[idx + 1];
    *sigma = f->sigma;
    memcpy(taps, f->taps, sizeof(float) * f->width);
    return f->width;
}

void orc_mesh(const orc_ctx *c, float *v, int *idx)
{
    int i, j, k;
    for (i = 0; i < ORC_NFACES; i++)
        for (j = 0; j < 3; j++) {
            for (k = 0; k < 3; k++)
                v[(i * 3 + j) * 3 + k] = c->mesh[i].v[j][k];
            idx[i * 3 + j] = c->mesh[i].idx[j];
        }
}

/* ---- Z-slab views of the window stages (multi-GPU driver tests) ------------ */
static void slab_level(orc_level_t *L, const float *data, int nx, int ny, int nz_loc,
                       int z_off, int nz_glob, const double *units, double sd)
{
    memset(L, 0, sizeof(*L));
    L->data = (float *)data;
    L->nx = nx; L->ny = ny; L->nz = nz_loc;
    L->ux = units[0]; L->uy = units[1]; L->uz = units[2];
    L->s = sd;
    L->z_off = z_off;
    L->nz_glob = nz_glob;
}

int orc_orient_slab(const float *data, int nx, int ny, int nz_loc, int z_off, int nz_glob,
                    const double *units, double sd, int x, int y, int z_glob,
                    double corner_thresh, float *R)
{
    orc_ctx c;
    orc_level_t L;
    const float ctr[3] = { (float)(double)x, (float)(double)y, (float)(double)z_glob };
    memset(&c, 0, sizeof(c));
    c.corner_thresh = corner_thresh;
    slab_level(&L, data, nx, ny, nz_loc, z_off, nz_glob, units, sd);
    return !orient_one(&c, &L, ctr, k_ori_sig_fctr * sd, R); /* 1 = kept */
}

void orc_describe_slab(const float *data, int nx, int ny, int nz_loc, int z_off, int nz_glob,
                       const double *units, const orc_keypoint *key, orc_descriptor *desc)
{
    static orc_ctx c;
    static int init = 0;
    orc_level_t L;
    if (!init) {
        build_mesh(c.mesh);
        init = 1;
    }
    slab_level(&L, data, nx, ny, nz_loc, z_off, nz_glob, units, key->sd);
    describe_one(&c, &L, key, desc);
}
