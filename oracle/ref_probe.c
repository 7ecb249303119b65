/* ref_probe.c -- TEST INFRASTRUCTURE ONLY (never linked into the product).
 *
 * Thin dump hooks around the *unmodified* reference sources.  The reference
 * translation units are pulled in where they lie under /root/reference (the
 * Makefile passes -I$(REF)/sift3d); nothing of theirs is copied into this
 * repository.  Including the .c files (rather than linking them) is what gives
 * this probe access to the reference's `static` stage functions
 * (set_im_SIFT3D, build_gpyr, build_dog, detect_extrema, assign_orientations,
 * convolve_sep, im_permute, init_Gauss_filter ...), so that every stage of the
 * hot path can be dumped separately for the golden vectors in tests/golden/.
 *
 * Everything below the two #includes is this repository's own code.
 */
#include "imutil.c" /* /root/reference/sift3d/imutil.c */
#include "sift.c"   /* /root/reference/sift3d/sift.c   */

#define PROBE __attribute__((visibility("default")))

/* ---- filters ------------------------------------------------------------ */

/* init_Gauss_filter (imutil.c:1267): returns width, writes taps. */
PROBE int probe_gauss_filter(double sigma, float *taps, int max_taps)
{
    sift3d_gauss_filter g;
    int w, i;
    if (init_Gauss_filter(&g, sigma, 3))
        return -1;
    w = g.f.width;
    for (i = 0; i < w && i < max_taps; i++)
        taps[i] = g.f.kernel[i];
    cleanup_Gauss_filter(&g);
    return w;
}

static void wrap_image(sift3d_image *im, float *data, int nx, int ny, int nz,
                       double ux, double uy, double uz)
{
    init_im(im);
    im->nx = nx;
    im->ny = ny;
    im->nz = nz;
    im->nc = 1;
    im_default_stride(im);
    im->size = (size_t)nx * ny * nz;
    im->data = data;
    im->ux = ux;
    im->uy = uy;
    im->uz = uz;
}

/* apply_Sep_FIR_filter (imutil.c:1127): all three axes. */
PROBE int probe_apply_sep_fir(const float *src, float *dst, int nx, int ny,
                              int nz, double ux, double uy, double uz,
                              const float *taps, int width, double unit)
{
    sift3d_image s, d;
    sift3d_sep_fir_filter f;
    int ret;
    wrap_image(&s, (float *)src, nx, ny, nz, ux, uy, uz);
    init_im(&d);
    f.kernel = (float *)taps;
    f.dim = 3;
    f.width = width;
    f.symmetric = 1;
    ret = apply_Sep_FIR_filter(&s, &d, &f, unit);
    if (!ret)
        memcpy(dst, d.data, (size_t)nx * ny * nz * sizeof(float));
    im_free(&d);
    return ret;
}

/* One axis only, through the same permute -> convolve(dim 0) -> permute
 * sequence apply_Sep_FIR_filter uses (imutil.c:1165-1188). */
PROBE int probe_fir_axis(const float *src, float *dst, int nx, int ny, int nz,
                         double ux, double uy, double uz, const float *taps,
                         int width, double unit, int axis)
{
    sift3d_image s, a, b, c;
    sift3d_sep_fir_filter f;
    int ret = SIFT3D_FAILURE;
    wrap_image(&s, (float *)src, nx, ny, nz, ux, uy, uz);
    init_im(&a);
    init_im(&b);
    init_im(&c);
    f.kernel = (float *)taps;
    f.dim = 3;
    f.width = width;
    f.symmetric = 1;
    if (axis == 0) {
        if (convolve_sep(&s, &c, &f, 0, unit))
            goto done;
    } else {
        if (im_permute(&s, 0, axis, &a) || convolve_sep(&a, &b, &f, 0, unit) ||
            im_permute(&b, 0, axis, &c))
            goto done;
    }
    memcpy(dst, c.data, (size_t)nx * ny * nz * sizeof(float));
    ret = SIFT3D_SUCCESS;
done:
    im_free(&a);
    im_free(&b);
    im_free(&c);
    return ret;
}

/* im_downsample_2x (imutil.c:591) */
PROBE int probe_downsample(const float *src, int nx, int ny, int nz, float *dst)
{
    sift3d_image s, d;
    wrap_image(&s, (float *)src, nx, ny, nz, 1, 1, 1);
    init_im(&d);
    if (im_downsample_2x(&s, &d))
        return -1;
    memcpy(dst, d.data, d.size * sizeof(float));
    im_free(&d);
    return 0;
}

/* eigen_Mat_rm (imutil.c:984): A row-major 3x3 -> Q (columns = eigenvectors),
 * L ascending. */
PROBE int probe_eigen3(const double *A9, double *Q9, double *L3)
{
    sift3d_mat_rm A, Q, L;
    int ret;
    if (init_Mat_rm(&A, 3, 3, SIFT3D_DOUBLE, SIFT3D_TRUE) ||
        init_Mat_rm(&Q, 0, 0, SIFT3D_DOUBLE, SIFT3D_TRUE) ||
        init_Mat_rm(&L, 0, 0, SIFT3D_DOUBLE, SIFT3D_TRUE))
        return -1;
    memcpy(A.u.data_double, A9, 9 * sizeof(double));
    ret = eigen_Mat_rm(&A, &Q, &L);
    if (!ret) {
        memcpy(Q9, Q.u.data_double, 9 * sizeof(double));
        memcpy(L3, L.u.data_double, 3 * sizeof(double));
    }
    cleanup_Mat_rm(&A);
    cleanup_Mat_rm(&Q);
    cleanup_Mat_rm(&L);
    return ret;
}

/* ---- staged detect -------------------------------------------------------- */

/* Candidate records captured between detect_extrema and assign_orientations. */
typedef struct {
    int o, s, x, y, z;
    float strength;
    double sd;
} probe_cand;

typedef struct {
    sift3d_detector *det;
    sift3d_keypoint_store *kp;
    sift3d_descriptor_store *desc;
    probe_cand *cand;
    int ncand;
} probe_ctx;

PROBE probe_ctx *probe_make(void)
{
    probe_ctx *c = (probe_ctx *)calloc(1, sizeof(*c));
    c->det = sift3d_make_detector();
    c->kp = sift3d_make_keypoint_store();
    c->desc = sift3d_make_descriptor_store();
    return c;
}

PROBE void probe_free(probe_ctx *c)
{
    sift3d_free_detector(c->det);
    sift3d_free_keypoint_store(c->kp);
    sift3d_free_descriptor_store(c->desc);
    free(c->cand);
    free(c);
}

PROBE sift3d_detector *probe_detector(probe_ctx *c) { return c->det; }

/* The body of sift3d_detect_keypoints (sift.c:1217-1249), stage by stage, with
 * the candidate list copied out before orientation assignment. */
PROBE int probe_detect(probe_ctx *c, const float *vol, int nx, int ny, int nz,
                       double ux, double uy, double uz)
{
    sift3d_image im;
    int i;
    wrap_image(&im, (float *)vol, nx, ny, nz, ux, uy, uz);
    if (set_im_SIFT3D(c->det, &im) || build_gpyr(c->det) || build_dog(c->det) ||
        detect_extrema(c->det, c->kp))
        return -1;
    c->ncand = (int)c->kp->slab.num;
    c->cand = (probe_cand *)realloc(c->cand, sizeof(probe_cand) * (c->ncand + 1));
    for (i = 0; i < c->ncand; i++) {
        const sift3d_keypoint *k = c->kp->buf + i;
        c->cand[i].o = k->o;
        c->cand[i].s = k->s;
        c->cand[i].x = (int)k->xd;
        c->cand[i].y = (int)k->yd;
        c->cand[i].z = (int)k->zd;
        c->cand[i].strength = k->strength;
        c->cand[i].sd = k->sd;
    }
    if (assign_orientations(c->det, c->kp))
        return -1;
    return 0;
}

/* The public entry point, for cross-checking the staged variant. */
PROBE int probe_detect_public(probe_ctx *c, const float *vol, int nx, int ny,
                              int nz)
{
    sift3d_image *im = sift3d_make_image(nx, ny, nz, 1);
    int ret;
    memcpy(sift3d_image_data(im), vol, (size_t)nx * ny * nz * sizeof(float));
    ret = sift3d_detect_keypoints(c->det, im, c->kp);
    sift3d_free_image(im);
    c->ncand = 0;
    return ret;
}

PROBE int probe_describe(probe_ctx *c)
{
    return sift3d_extract_descriptors(c->det, c->kp, c->desc);
}

PROBE void probe_sort(probe_ctx *c, int limit)
{
    sift3d_keypoint_store_sort_by_strength(c->kp, limit);
}

PROBE int probe_num_octaves(probe_ctx *c) { return c->det->gpyr.num_octaves; }
PROBE int probe_num_cand(probe_ctx *c) { return c->ncand; }
PROBE int probe_num_kp(probe_ctx *c) { return (int)c->kp->slab.num; }

PROBE void probe_get_cand(probe_ctx *c, int *osxyz, float *strength, double *sd)
{
    int i;
    for (i = 0; i < c->ncand; i++) {
        osxyz[5 * i + 0] = c->cand[i].o;
        osxyz[5 * i + 1] = c->cand[i].s;
        osxyz[5 * i + 2] = c->cand[i].x;
        osxyz[5 * i + 3] = c->cand[i].y;
        osxyz[5 * i + 4] = c->cand[i].z;
        strength[i] = c->cand[i].strength;
        sd[i] = c->cand[i].sd;
    }
}

PROBE void probe_get_kp(probe_ctx *c, int *os, double *xyzsd, float *strength,
                        float *R)
{
    int i, n = (int)c->kp->slab.num;
    for (i = 0; i < n; i++) {
        const sift3d_keypoint *k = c->kp->buf + i;
        os[2 * i] = k->o;
        os[2 * i + 1] = k->s;
        xyzsd[4 * i] = k->xd;
        xyzsd[4 * i + 1] = k->yd;
        xyzsd[4 * i + 2] = k->zd;
        xyzsd[4 * i + 3] = k->sd;
        strength[i] = k->strength;
        memcpy(R + 9 * i, k->R.u.data_float, 9 * sizeof(float));
    }
}

/* which: 0 = Gaussian pyramid, 1 = DoG pyramid, 2 = scaled input copy */
PROBE const float *probe_level(probe_ctx *c, int which, int o, int s, int *dims,
                               double *units, double *scale)
{
    const sift3d_image *im;
    if (which == 2)
        im = &c->det->im;
    else
        im = SIFT3D_PYR_IM_GET(which ? &c->det->dog : &c->det->gpyr, o, s);
    dims[0] = im->nx;
    dims[1] = im->ny;
    dims[2] = im->nz;
    units[0] = im->ux;
    units[1] = im->uy;
    units[2] = im->uz;
    *scale = im->s;
    return im->data;
}

/* Filter bank of the detector after an image was set (make_gss, imutil.c:1360).
 * idx = -1: first_gauss; idx >= 0: gauss_octave[idx]. */
PROBE int probe_gss(probe_ctx *c, int idx, double *sigma, float *taps)
{
    const sift3d_gauss_filter *g = idx < 0 ? &c->det->gss.first_gauss
                                           : c->det->gss.gauss_octave + idx;
    if (idx >= c->det->gss.num_filters)
        return -1;
    *sigma = g->sigma;
    memcpy(taps, g->f.kernel, g->f.width * sizeof(float));
    return g->f.width;
}

/* Icosahedron table built by init_geometry (sift.c:148). */
PROBE void probe_mesh(probe_ctx *c, float *v /*20*3*3*/, int *idx /*20*3*/)
{
    int i, j;
    for (i = 0; i < ICOS_NFACES; i++)
        for (j = 0; j < 3; j++) {
            v[(i * 3 + j) * 3 + 0] = c->det->mesh.tri[i].v[j].x;
            v[(i * 3 + j) * 3 + 1] = c->det->mesh.tri[i].v[j].y;
            v[(i * 3 + j) * 3 + 2] = c->det->mesh.tri[i].v[j].z;
            idx[i * 3 + j] = c->det->mesh.tri[i].idx[j];
        }
}

/* Raw descriptor records: 768 floats + xd,yd,zd,sd. */
PROBE int probe_get_desc(probe_ctx *c, float *hist, double *xyzsd)
{
    int i, n = (int)c->desc->num;
    for (i = 0; i < n; i++) {
        memcpy(hist + (size_t)i * DESC_NUMEL, c->desc->buf[i].hists,
               DESC_NUMEL * sizeof(float));
        xyzsd[4 * i] = c->desc->buf[i].xd;
        xyzsd[4 * i + 1] = c->desc->buf[i].yd;
        xyzsd[4 * i + 2] = c->desc->buf[i].zd;
        xyzsd[4 * i + 3] = c->desc->buf[i].sd;
    }
    return n;
}

/* Public converters, for the mat_rm layouts (sift.c:1644-1726). */
PROBE int probe_kp_mat(probe_ctx *c, double *out)
{
    sift3d_mat_rm *m = sift3d_make_mat_rm();
    int rows = 0, cols = 0;
    if (sift3d_keypoint_store_to_mat_rm(c->kp, m))
        return -1;
    sift3d_mat_rm_dimensions(m, &cols, &rows);
    memcpy(out, sift3d_mat_rm_data(m), sizeof(double) * rows * cols);
    sift3d_free_mat_rm(m);
    return rows;
}

PROBE int probe_desc_mat(probe_ctx *c, float *out)
{
    sift3d_mat_rm *m = sift3d_make_mat_rm();
    int rows = 0, cols = 0;
    if (sift3d_descriptor_store_to_mat_rm(c->desc, m))
        return -1;
    sift3d_mat_rm_dimensions(m, &cols, &rows);
    memcpy(out, sift3d_mat_rm_data(m), sizeof(float) * rows * cols);
    sift3d_free_mat_rm(m);
    return rows;
}

PROBE int probe_save(probe_ctx *c, const char *kp_path, const char *desc_path)
{
    int r = 0;
    if (kp_path)
        r |= sift3d_keypoint_store_save(kp_path, c->kp);
    if (desc_path)
        r |= sift3d_descriptor_store_save(desc_path, c->desc);
    return r;
}

/* Stage timing of detect (for DESIGN.md's "literal reference" CPU numbers). */
#include <time.h>
static double now_s(void)
{
    struct timespec t;
    clock_gettime(CLOCK_MONOTONIC, &t);
    return t.tv_sec + 1e-9 * t.tv_nsec;
}

PROBE int probe_time_detect(probe_ctx *c, const float *vol, int nx, int ny,
                            int nz, double *t /*5: set_im,gpyr,dog,extrema,orient*/)
{
    sift3d_image im;
    double t0;
    wrap_image(&im, (float *)vol, nx, ny, nz, 1, 1, 1);
    t0 = now_s();
    if (set_im_SIFT3D(c->det, &im))
        return -1;
    t[0] = now_s() - t0;
    t0 = now_s();
    if (build_gpyr(c->det))
        return -1;
    t[1] = now_s() - t0;
    t0 = now_s();
    if (build_dog(c->det))
        return -1;
    t[2] = now_s() - t0;
    t0 = now_s();
    if (detect_extrema(c->det, c->kp))
        return -1;
    t[3] = now_s() - t0;
    c->ncand = (int)c->kp->slab.num;
    t0 = now_s();
    if (assign_orientations(c->det, c->kp))
        return -1;
    t[4] = now_s() - t0;
    return 0;
}
