/* sift3d_host.c -- C host side of the MI355X drop-in for fatimp/SIFT3D v2.0.
 *
 * Implements the 27 public symbols of the reference library (the headers under include/sift3d/,
 * reference: sift3d/sift.h, sift3d/imutil.h) with the reference's object semantics,
 * parameter checks and error behaviour, and drives the detect / describe hot path
 * through the device-level C ABI of include/sift3d_amd.h (sift3d_kernels.hip).  There
 * is NO CPU fallback: without a HIP device the two hot entry points fail loudly.
 *
 * What lives where:
 *   host   object lifetimes, parameter validation, octave/level geometry, the Gaussian
 *          filter bank (computed with the host libm exactly as the reference does,
 *          imutil.c:1267-1343), candidate -> keypoint compaction (with the reference's
 *          stale-strength quirk), stores, converters, CSV writers
 *   HBM    the scaled input, both pyramids, scratch volumes, candidate / keypoint /
 *          descriptor records -- resident across sift3d_detect_keypoints and
 *          sift3d_extract_descriptors like the reference's retained pyramids
 *          (sift.c:1544-1549)
 *
 * Citations are file:line under /root/reference/sift3d/.
 */
#include <float.h>
#include <math.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <time.h>
#include <zlib.h>
#include <omp.h>
#include <dlfcn.h>
#include <pthread.h>

#include "../../include/sift3d/imutil.h"
#include "../../include/sift3d/sift.h"
#include "../../include/sift3d_amd.h"

#define ERR(...) fprintf(stderr, __VA_ARGS__) /* SIFT3D_ERR, immacros.h:31 */

#define NFACES 20
#define NVERT 12
#define DESC_NUMEL 768
#define SLAB_LEN 500 /* SIFT3D_SLAB_LEN, immacros.h:199-201 */

/* sift.c:31-35, :48 */
static const double peak_thresh_default = 0.1;
static const int num_kp_levels_default = 3;
static const double corner_thresh_default = 0.4;
static const double sigma_n_default = 1.15;
static const double sigma0_default = 1.6;
static const double golden_ratio = 1.6180339887;

/* ------------------------------------------------------------------------ */
/* object layouts (private; the API only exposes opaque handles)             */
/* ------------------------------------------------------------------------ */
struct _sift3d_image {
    float *data;
    size_t size;
    int nx, ny, nz, nc;
    double ux, uy, uz;
    int pinned;            /* data is page-locked (sift3d_hip_host_alloc) */
};

struct _sift3d_mat_rm {
    void *data;
    size_t size; /* bytes */
    int num_cols, num_rows;
    sift3d_mat_type type;
};

typedef struct {
    float R[9];
    double xd, yd, zd, sd;
    int o, s;
    float strength;
} keypoint_t;

struct _sift3d_keypoint_store {
    keypoint_t *buf;
    size_t num, cap;
    int nx, ny, nz;
};

/* Descriptors are kept as two arrays: the histograms (num x 768 floats, page-locked so that
 * they are the direct target of the device-to-host copy) and the coordinates. */
struct _sift3d_descriptor_store {
    float *hist;       /* [num][768]: cell = cx + 4 cy + 16 cz, 12 vertex bins each */
    double *xyzsd;     /* [num][4]: xd, yd, zd (octave-0 voxels), sd */
    size_t num, cap;
    int pinned;
    /* optional copy of the histograms in device memory (sift3d_amd_descriptor_store_keep_device) */
    int keep_device;
    float *d_hist;
    size_t d_cap, d_num;   /* d_num == num && d_num > 0: the copy is current */
    int nx, ny, nz;
};

typedef struct {
    double sigma;
    int width;
    float *taps;
} filter_t;

struct _sift3d_detector {
    /* parameters (sift.c:499-565) */
    double peak_thresh, corner_thresh, sigma_n, sigma0;
    int cuboid_extrema;     /* 0: 8-neighbour test (default build), 1: CUBOID_EXTREMA, sift.c:24 */
    int num_kp_levels;
    /* image geometry */
    int have_im;
    int nx, ny, nz;
    double units[3];       /* units of the current image              */
    double alloc_units[3]; /* units of the image that sized the pyramid */
    /* pyramid geometry */
    int num_octaves, ngl, ndl;
    int (*odims)[3];       /* per octave */
    filter_t *filt;        /* [0] first blur, [1..ngl-1] octave filters */
    int nfilt;
    /* device state */
    void *stream, *copy_stream;
    void *oct_stream;      /* octaves >= 1 of the pyramid, beside the last levels of octave 0 */
    void *side_stream;     /* ... and their levels that no later octave depends on */
    void *ev_fork, *ev_join, *ev_join2, *ev_part;
    int parts_timed;       /* the last detect oriented octave 0 on its own (ev_part, ev_join: the parts' ends) */
    void *ev_blur[SIFT3D_AMD_TIMED_BLURS][3]; /* octave 0, blur s: before its x pass, between x and the fused
                            * y+z launch, after it (on the stream they run on) */
    unsigned yz_timed;     /* bit s: blur s of octave 0 took the fused y+z kernel in the last detect */
    void *ev_pyr[2];       /* the pyramid's last launch on the octave stream / the side stream */
    int pyr_chains;        /* the last detect built its pyramid on three chains (ev_pyr are recorded) */
    void *ev_oct[32];      /* per octave: its downsampling source level is complete */
    int device;            /* HIP device of the streams / pyramids */
    void *ev[8];
    void *ev_chunk[8];
    float *d_im, *d_tmp_a, *d_tmp_b, *d_in;
    float *d_tmp2_a, *d_tmp2_b;   /* scratch volumes of the octave stream (octave-1 size) */
    float *d_tmp3_a, *d_tmp3_b;   /* ... and of the side stream */
    size_t in_cap;
    float **d_g, **d_d;    /* [num_octaves*ngl], [num_octaves*ndl] (DoG: only where stored) */
    unsigned char dog_free[64]; /* per octave: the last detect formed its DoG levels on the fly */
    float *d_scalars;      /* [0] input max, [1] count (as u32), [8 + o*ndl + s] dogmax, then
                            * [8 + (num_octaves + o)*ndl + s] their lower bounds (sift3d_hip_dogmax_sub) */
    int t_pending;         /* stage events not yet read into t[]: 1 detect, 2 describe */
    int orient_serial;     /* sift3d_amd_detector_set_serial_orientation */
    int exact_desc;        /* sift3d_amd_detector_set_exact_descriptors: 0 auto, 1 always, -1 never */
    int im_valid;          /* d_im holds the scaled image of the last detect call (else: see last_vol) */
    const float *last_vol; /* the last detect call's volume on the device (the caller's, or d_in) */
    int est0;              /* the large octaves' maxima gathered by their extrema sweeps (default; 0: a pass
                            * of their own) */
    sift3d_hip_level *h_levels, *d_levels;
    sift3d_hip_cand *d_cand, *h_cand;
    uint32_t cand_cap;
    float *h_R;            /* page-locked, device-visible: the orientation kernels write here */
    int32_t *h_keep;
    void *d_work;
    size_t work_bytes;
    void *d_work2;         /* extrema work areas of octaves >= 1 (kept between the two phases) */
    size_t work2_off[64], work2_bytes;
    float *d_wlut;         /* per-level window-weight tables of the descriptor kernel */
    void *d_dpart;         /* ... and the scratch of its split windows (sift3d_hip_describe_parts) */
    size_t dpart_bytes;
    void *d_otab;          /* window tables + per-candidate sums of the orientation kernels */
    size_t otab_bytes;
    sift3d_hip_kp *h_kp;    /* the describe kernel's input list (page-locked; read by the kernel in place) */
    uint32_t kp_cap;
    int have_pyramid;
    int ncand;
    double t[SIFT3D_AMD_NUM_TIMINGS];
};

static pthread_mutex_t g_mesh_lock = PTHREAD_MUTEX_INITIALIZER;
static unsigned char g_mesh_ready[64];   /* per device: the __constant__ tables live on ONE device */

static double now_s(void)
{
    struct timespec t;
    clock_gettime(CLOCK_MONOTONIC, &t);
    return t.tv_sec + 1e-9 * t.tv_nsec;
}

static const char k_version[] = "sift3d_amd 0.1 (gfx950)";
const char *sift3d_amd_version(void) { return k_version; }

/* ------------------------------------------------------------------------ */
/* images (imutil.c:1639-1674)                                               */
/* ------------------------------------------------------------------------ */
sift3d_image *sift3d_make_image(const int nx, const int ny, const int nz, const int nc)
{
    sift3d_image *im;
    /* im_resize, imutil.c:560-575 */
    if (nx <= 0 || ny <= 0 || nz <= 0) {
        ERR("im_resize: invalid dimension: %d x %d x %d \n", nx, ny, nz);
        return NULL;
    }
    if (nc < 1) {
        ERR("im_resize: invalid number of channels: %d \n", nc);
        return NULL;
    }
    im = (sift3d_image *)calloc(1, sizeof(*im));
    if (!im)
        return NULL;
    im->nx = nx; im->ny = ny; im->nz = nz; im->nc = nc;
    im->ux = im->uy = im->uz = 1;               /* init_im, imutil.c:1249-1251 */
    im->size = (size_t)nx * ny * nz * nc;
    /* The raster is what sift3d_detect_keypoints uploads: page-locked where a device is there to upload to
     * (the copy then runs at the link's rate instead of through the runtime's staging buffers: 512 MB in
     * ~9 instead of ~12 ms), plain memory otherwise.  Zeroed either way (im_zero, imutil.c:507). */
    if (im->size * sizeof(float) >= ((size_t)1 << 20) && sift3d_amd_device_available() &&
        (im->data = (float *)sift3d_hip_host_alloc(im->size * sizeof(float)))) {
        im->pinned = 1;
        memset(im->data, 0, im->size * sizeof(float));
    } else {
        im->data = (float *)calloc(im->size, sizeof(float));
    }
    if (!im->data) {
        free(im);
        return NULL;
    }
    return im;
}

void sift3d_free_image(sift3d_image *im)
{
    if (!im)
        return;
    if (im->pinned)
        sift3d_hip_host_free(im->data);
    else
        free(im->data);
    free(im);
}

static const char *file_ext(const char *path)
{
    /* get_file_ext, imutil.c:303-315 */
    const char *name = strrchr(path, '/');
    const char *dot;
    name = name ? name : path;
    dot = strrchr(name, '.');
    return (!dot || dot == name) ? "" : dot + 1;
}

/* ---- NIfTI-1 single-file reader (.nii, .nii.gz) ---------------------------------------
 * The reference reads images through nifticlib (nifti.c:52-167); that library is not part of
 * this build, so the subset the reference actually uses is read directly: header fields dim,
 * datatype, pixdim, vox_offset, scl_slope, scl_inter; either byte order; the ten integer/float
 * sample types of nifti.c:118-148; value = (float)((double)raw * slope + inter) with slope 0
 * read as 1 (nifti.c:107-116); a 4th dimension becomes channels stored innermost
 * (im_default_stride, nifti.c:44-46).  Analyze/NIfTI pairs (.img/.hdr) are not read. */
static void swap_bytes(void *p, size_t size, size_t count)
{
    unsigned char *b = (unsigned char *)p;
    size_t i, j;
    for (i = 0; i < count; i++, b += size)
        for (j = 0; j < size / 2; j++) {
            const unsigned char t = b[j];
            b[j] = b[size - 1 - j];
            b[size - 1 - j] = t;
        }
}

static double nii_sample(const unsigned char *p, int datatype)
{
    switch (datatype) {
    case 2: return (double)*(const uint8_t *)p;
    case 256: return (double)*(const int8_t *)p;
    case 4: { int16_t v; memcpy(&v, p, 2); return (double)v; }
    case 512: { uint16_t v; memcpy(&v, p, 2); return (double)v; }
    case 8: { int32_t v; memcpy(&v, p, 4); return (double)v; }
    case 768: { uint32_t v; memcpy(&v, p, 4); return (double)v; }
    case 1024: { int64_t v; memcpy(&v, p, 8); return (double)v; }
    case 1280: { uint64_t v; memcpy(&v, p, 8); return (double)v; }
    case 16: { float v; memcpy(&v, p, 4); return (double)v; }
    default: { double v; memcpy(&v, p, 8); return v; }           /* 64 */
    }
}

static sift3d_image *read_nii(const char *path)
{
    unsigned char hdr[348];
    int16_t dim[8], datatype;
    float pixdim[8], vox_offset, slope_f, inter_f;
    int32_t sizeof_hdr;
    int swap, i, ndim, dim_counter, nx, ny, nz, nc, x, y, z, c;
    size_t bytes_per, nvox, row_bytes;
    double slope;
    unsigned char *row = NULL;
    sift3d_image *im = NULL;
    gzFile f = gzopen(path, "rb");                 /* reads plain files too */
    if (!f) {
        ERR("read_nii: failure loading file %s", path);        /* nifti.c:63 */
        return NULL;
    }
    if (gzread(f, hdr, 348) != 348)
        goto bad_file;
    memcpy(&sizeof_hdr, hdr, 4);
    swap = sizeof_hdr != 348;
    if (swap) {
        swap_bytes(&sizeof_hdr, 4, 1);
        if (sizeof_hdr != 348)
            goto bad_file;
    }
    if (memcmp(hdr + 344, "n+1", 4)) {             /* "ni1": header/image pair */
        ERR("read_nii: %s is not a single-file NIFTI-1 image \n", path);
        goto fail;
    }
    memcpy(dim, hdr + 40, 16);
    memcpy(&datatype, hdr + 70, 2);
    memcpy(pixdim, hdr + 76, 32);
    memcpy(&vox_offset, hdr + 108, 4);
    memcpy(&slope_f, hdr + 112, 4);
    memcpy(&inter_f, hdr + 116, 4);
    if (swap) {
        swap_bytes(dim, 2, 8);
        swap_bytes(&datatype, 2, 1);
        swap_bytes(pixdim, 4, 8);
        swap_bytes(&vox_offset, 4, 1);
        swap_bytes(&slope_f, 4, 1);
        swap_bytes(&inter_f, 4, 1);
    }
    ndim = dim[0];
    if (ndim < 1 || ndim > 7)
        goto bad_file;
    for (i = 1; i <= ndim; i++)
        if (dim[i] < 1)
            goto bad_file;
    /* dimensionality = last dimension greater than 1 (nifti.c:67-72) */
    for (dim_counter = ndim; dim_counter > 0; dim_counter--)
        if (dim[dim_counter] > 1)
            break;
    if (dim_counter > 4) {
        ERR("read_nii: file %s has unsupported dimensionality %d\n", path, dim_counter);
        goto fail;
    }
    nx = dim[1];
    ny = ndim >= 2 ? dim[2] : 1;
    nz = ndim >= 3 ? dim[3] : 1;
    nc = dim_counter == 4 ? dim[4] : 1;            /* nifti.c:97 */
    switch (datatype) {
    case 2: case 256: bytes_per = 1; break;
    case 4: case 512: bytes_per = 2; break;
    case 8: case 768: case 16: bytes_per = 4; break;
    case 1024: case 1280: case 64: bytes_per = 8; break;
    default:
        ERR("read_nii: unsupported datatype %d \n", (int)datatype);     /* nifti.c:149-155 */
        goto fail;
    }
    if (!(vox_offset >= 348.0f) || gzseek(f, (z_off_t)vox_offset, SEEK_SET) < 0)
        goto bad_file;
    if (!(im = sift3d_make_image(nx, ny, nz, nc)))
        goto fail;
    /* real world coordinates, nifti.c:87-90 (nifticlib: dx,dy,dz = pixdim[1..3]) */
    im->ux = pixdim[1];
    im->uy = pixdim[2];
    im->uz = pixdim[3];
    if (!(im->ux > 0) || !(im->uy > 0) || !(im->uz > 0)) {
        ERR("read_nii: file %s has a non-positive voxel spacing (%f, %f, %f) \n", path, im->ux,
            im->uy, im->uz);
        goto fail;
    }
    slope = slope_f;
    if (slope == 0.0)
        slope = 1.0;                               /* nifti.c:107-109 */
    nvox = (size_t)nx * ny * nz;
    row_bytes = (size_t)nx * bytes_per;
    if (!(row = (unsigned char *)malloc(row_bytes)))
        goto fail;
    for (c = 0; c < nc; c++)                       /* file order: channel slowest */
        for (z = 0; z < nz; z++)
            for (y = 0; y < ny; y++) {
                if ((size_t)gzread(f, row, (unsigned)row_bytes) != row_bytes)
                    goto bad_file;
                if (swap)
                    swap_bytes(row, bytes_per, (size_t)nx);
                for (x = 0; x < nx; x++)
                    im->data[(size_t)c + (size_t)nc * ((size_t)x + (size_t)nx * ((size_t)y + (size_t)ny * z))] =
                        (float)(nii_sample(row + (size_t)x * bytes_per, datatype) * slope +
                                (double)inter_f);                      /* nifti.c:112-116 */
            }
    (void)nvox;
    free(row);
    gzclose(f);
    return im;
bad_file:
    ERR("read_nii: failure loading file %s", path);
fail:
    free(row);
    sift3d_free_image(im);
    gzclose(f);
    return NULL;
}

sift3d_image *sift3d_read_image(const char *path)
{
    const char *ext = file_ext(path);
    if (!strcmp(ext, "gz") || !strcmp(ext, "nii"))             /* im_get_format, imutil.c:318-333 */
        return read_nii(path);
    if (!strcmp(ext, "img")) {
        /* Analyze pairs went through nifticlib (nifti.c:16-31: not part of this build) */
        ERR("sift3d_read_image: Analyze (.img/.hdr) images are not supported by this build; use "
            "single-file NIFTI (.nii, .nii.gz) or fill an image made with sift3d_make_image() \n");
    } else {
        ERR("im_read: unrecognized file extension from file %s \n", path); /* imutil.c:366 */
    }
    return NULL;
}

float *sift3d_image_data(const sift3d_image *im) { return im->data; }

int sift3d_amd_image_info(const sift3d_image *im, int *dims4, double *units3)
{
    if (!im)
        return SIFT3D_FAILURE;
    if (dims4) {
        dims4[0] = im->nx; dims4[1] = im->ny; dims4[2] = im->nz; dims4[3] = im->nc;
    }
    if (units3) {
        units3[0] = im->ux; units3[1] = im->uy; units3[2] = im->uz;
    }
    return SIFT3D_SUCCESS;
}

int sift3d_amd_image_set_units(sift3d_image *im, double ux, double uy, double uz)
{
    if (!im || !(ux > 0) || !(uy > 0) || !(uz > 0))
        return SIFT3D_FAILURE;
    im->ux = ux; im->uy = uy; im->uz = uz;
    return SIFT3D_SUCCESS;
}

/* ------------------------------------------------------------------------ */
/* matrices (imutil.c:226-279, 1676-1710)                                    */
/* ------------------------------------------------------------------------ */
static int mat_resize(sift3d_mat_rm *m, int rows, int cols, sift3d_mat_type type)
{
    const size_t el = type == SIFT3D_DOUBLE ? sizeof(double)
                                            : type == SIFT3D_FLOAT ? sizeof(float) : sizeof(int);
    const size_t total = el * (size_t)rows * (size_t)cols;
    m->num_rows = rows;
    m->num_cols = cols;
    m->type = type;
    if (total == m->size)
        return SIFT3D_SUCCESS;
    if (total == 0) {
        free(m->data);
        m->data = NULL;
        m->size = 0;
        return SIFT3D_SUCCESS;
    }
    {
        void *p = realloc(m->data, total);
        if (!p) {
            free(m->data);
            m->data = NULL;
            m->size = 0;
            return SIFT3D_FAILURE;
        }
        m->data = p;
        m->size = total;
    }
    return SIFT3D_SUCCESS;
}

sift3d_mat_rm *sift3d_make_mat_rm()
{
    sift3d_mat_rm *m = (sift3d_mat_rm *)calloc(1, sizeof(*m));
    if (m)
        m->type = SIFT3D_FLOAT; /* imutil.c:1678 */
    return m;
}

void sift3d_free_mat_rm(sift3d_mat_rm *m)
{
    if (!m)
        return;
    free(m->data);
    free(m);
}

void *sift3d_mat_rm_data(sift3d_mat_rm *m) { return m->data; }

void sift3d_mat_rm_dimensions(const sift3d_mat_rm *m, int *num_cols, int *num_rows)
{
    if (num_cols)
        *num_cols = m->num_cols;
    if (num_rows)
        *num_rows = m->num_rows;
}

sift3d_mat_type sift3d_mat_rm_type(const sift3d_mat_rm *m) { return m->type; }

/* write_Mat_rm, imutil.c:405-479: "%f" / "%d", ',' between columns, '\n' after a row,
 * gzip when the extension is "gz" */
static int mat_write(const char *path, const sift3d_mat_rm *m)
{
    const int compress = strcmp(file_ext(path), "gz") == 0;
    FILE *f = NULL;
    gzFile gz = NULL;
    int i, j, ok = 1;
    char buf[64];
    if (compress) {
        if (!(gz = gzopen(path, "w")))
            return SIFT3D_FAILURE;
    } else if (!(f = fopen(path, "w"))) {
        return SIFT3D_FAILURE;
    }
    for (i = 0; i < m->num_rows && ok; i++)
        for (j = 0; j < m->num_cols; j++) {
            const size_t k = (size_t)j + (size_t)i * m->num_cols;
            const char delim = j < m->num_cols - 1 ? ',' : '\n';
            int len;
            switch (m->type) {
            case SIFT3D_DOUBLE: len = snprintf(buf, sizeof(buf), "%f", ((double *)m->data)[k]); break;
            case SIFT3D_FLOAT: len = snprintf(buf, sizeof(buf), "%f", ((float *)m->data)[k]); break;
            default: len = snprintf(buf, sizeof(buf), "%d", ((int *)m->data)[k]); break;
            }
            if (len >= (int)sizeof(buf) - 1) { /* huge magnitudes: fall back to direct printf */
                if (compress)
                    gzprintf(gz, "%f", m->type == SIFT3D_DOUBLE ? ((double *)m->data)[k]
                                                                  : (double)((float *)m->data)[k]);
                else
                    fprintf(f, "%f", m->type == SIFT3D_DOUBLE ? ((double *)m->data)[k]
                                                                : (double)((float *)m->data)[k]);
                len = 0;
            }
            buf[len] = delim;
            if (compress) {
                if (gzwrite(gz, buf, (unsigned)len + 1) != len + 1)
                    ok = 0;
            } else if (fwrite(buf, 1, (size_t)len + 1, f) != (size_t)len + 1) {
                ok = 0;
            }
        }
    if (compress) {
        if (gzclose(gz) != Z_OK)
            ok = 0;
    } else {
        if (ferror(f))
            ok = 0;
        fclose(f);
    }
    return ok ? SIFT3D_SUCCESS : SIFT3D_FAILURE;
}

/* ------------------------------------------------------------------------ */
/* stores (sift.c:329-423, 1861-1900)                                        */
/* ------------------------------------------------------------------------ */
sift3d_keypoint_store *sift3d_make_keypoint_store()
{
    return (sift3d_keypoint_store *)calloc(1, sizeof(sift3d_keypoint_store));
}

void sift3d_free_keypoint_store(sift3d_keypoint_store *kp)
{
    if (!kp)
        return;
    free(kp->buf);
    free(kp);
}

/* resize_Keypoint_store: capacity moves in slabs of 500 records (immacros.h:202-222) */
static int kp_store_resize(sift3d_keypoint_store *kp, size_t num)
{
    const size_t cap = ((num + SLAB_LEN - 1) / SLAB_LEN) * SLAB_LEN;
    if (cap != kp->cap) {
        if (cap == 0) {
            free(kp->buf);
            kp->buf = NULL;
        } else {
            keypoint_t *p = (keypoint_t *)realloc(kp->buf, cap * sizeof(keypoint_t));
            if (!p) {
                free(kp->buf);
                kp->buf = NULL;
                kp->cap = kp->num = 0;
                return SIFT3D_FAILURE;
            }
            kp->buf = p;
        }
        kp->cap = cap;
    }
    kp->num = num;
    return SIFT3D_SUCCESS;
}

sift3d_descriptor_store *sift3d_make_descriptor_store()
{
    return (sift3d_descriptor_store *)calloc(1, sizeof(sift3d_descriptor_store));
}

static void desc_store_release(sift3d_descriptor_store *d)
{
    sift3d_hip_free(d->d_hist);
    d->d_hist = NULL;
    d->d_cap = d->d_num = 0;
    if (d->pinned)
        sift3d_hip_host_free(d->hist);
    else
        free(d->hist);
    free(d->xyzsd);
    d->hist = NULL;
    d->xyzsd = NULL;
    d->num = d->cap = 0;
    d->pinned = 0;
}

void sift3d_free_descriptor_store(sift3d_descriptor_store *d)
{
    if (!d)
        return;
    desc_store_release(d);
    free(d);
}

/* keypoint_strength_cmp never returns 0 (sift.c:1832-1837, quirk Q7) */
static int strength_cmp(const void *a, const void *b)
{
    return (((const keypoint_t *)a)->strength < ((const keypoint_t *)b)->strength) ? 1 : -1;
}

void sift3d_keypoint_store_sort_by_strength(sift3d_keypoint_store *const store, int limit)
{
    if (!store->num)
        return;
    qsort(store->buf, store->num, sizeof(keypoint_t), strength_cmp);
    if (store->num > (size_t)limit && limit != 0)  /* sift.c:1897-1899 */
        kp_store_resize(store, (size_t)limit);
}

int sift3d_keypoint_store_to_mat_rm(const sift3d_keypoint_store *const kp, sift3d_mat_rm *const mat)
{
    const int num = (int)kp->num;
    int i;
    if (mat_resize(mat, num, 3, SIFT3D_DOUBLE))
        return SIFT3D_FAILURE;
    for (i = 0; i < num; i++) {
        const keypoint_t *k = kp->buf + i;
        const double f = ldexp(1.0, k->o);        /* sift.c:1663 */
        double *row = (double *)mat->data + 3 * (size_t)i;
        row[0] = f * k->xd;
        row[1] = f * k->yd;
        row[2] = f * k->zd;
    }
    return SIFT3D_SUCCESS;
}

int sift3d_descriptor_store_to_mat_rm(const sift3d_descriptor_store *const store,
                                      sift3d_mat_rm *const mat)
{
    const int rows = (int)store->num, cols = 3 + DESC_NUMEL;
    int i;
    if (rows < 1) {                                /* sift.c:1691-1695 */
        printf("SIFT3D_Descriptor_store_to_Mat_rm: invalid number of descriptors: %d \n", rows);
        return SIFT3D_FAILURE;
    }
    if (mat_resize(mat, rows, cols, SIFT3D_FLOAT))
        return SIFT3D_FAILURE;
    for (i = 0; i < rows; i++) {
        float *row = (float *)mat->data + (size_t)cols * i;
        row[0] = (float)store->xyzsd[4 * (size_t)i];
        row[1] = (float)store->xyzsd[4 * (size_t)i + 1];
        row[2] = (float)store->xyzsd[4 * (size_t)i + 2];
        memcpy(row + 3, store->hist + (size_t)DESC_NUMEL * i,
               sizeof(float) * DESC_NUMEL);            /* col = 3 + 12*cell + bin */
    }
    return SIFT3D_SUCCESS;
}

int sift3d_keypoint_store_save(const char *path, const sift3d_keypoint_store *const kp)
{
    /* columns: strength, x, y, z, o, sd, R00..R22 (sift.c:1746-1789) */
    sift3d_mat_rm m;
    const int rows = (int)kp->num, cols = 15;
    int i, j, ret;
    memset(&m, 0, sizeof(m));
    if (mat_resize(&m, rows, cols, SIFT3D_DOUBLE))
        return SIFT3D_FAILURE;
    for (i = 0; i < rows; i++) {
        const keypoint_t *k = kp->buf + i;
        double *row = (double *)m.data + (size_t)cols * i;
        row[0] = k->strength;
        row[1] = k->xd;
        row[2] = k->yd;
        row[3] = k->zd;
        row[4] = k->o;
        row[5] = k->sd;
        for (j = 0; j < 9; j++)
            row[6 + j] = (double)k->R[j];
    }
    ret = mat_write(path, &m);
    free(m.data);
    return ret;
}

int sift3d_descriptor_store_save(const char *path, const sift3d_descriptor_store *const desc)
{
    sift3d_mat_rm m;
    int ret;
    memset(&m, 0, sizeof(m));
    m.type = SIFT3D_FLOAT;
    if (sift3d_descriptor_store_to_mat_rm(desc, &m)) {
        free(m.data);
        return SIFT3D_FAILURE;
    }
    ret = mat_write(path, &m);
    free(m.data);
    return ret;
}

int sift3d_amd_keypoint_store_size(const sift3d_keypoint_store *kp) { return (int)kp->num; }
int sift3d_amd_descriptor_store_size(const sift3d_descriptor_store *d) { return (int)d->num; }

/* Fill a descriptor store from host arrays (tests of the writers / converters without a
 * device): n records of {x, y, z, sd} (doubles) and 768 floats. */
int sift3d_amd_descriptor_store_set(sift3d_descriptor_store *d, int n, const double *xyz_sd,
                                    const float *hist, int nx, int ny, int nz)
{
    if (!d || n < 0 || (n && (!xyz_sd || !hist)))
        return SIFT3D_FAILURE;
    desc_store_release(d);
    if (n) {
        d->hist = (float *)malloc(sizeof(float) * DESC_NUMEL * (size_t)n);
        d->xyzsd = (double *)malloc(sizeof(double) * 4 * (size_t)n);
        if (!d->hist || !d->xyzsd) {
            desc_store_release(d);
            return SIFT3D_FAILURE;
        }
        memcpy(d->hist, hist, sizeof(float) * DESC_NUMEL * (size_t)n);
        memcpy(d->xyzsd, xyz_sd, sizeof(double) * 4 * (size_t)n);
    }
    d->cap = d->num = (size_t)n;
    d->nx = nx; d->ny = ny; d->nz = nz;
    return SIFT3D_SUCCESS;
}

int sift3d_amd_keypoint_store_get(const sift3d_keypoint_store *kp, int i, int *o, int *s,
                                  double *xyz_sd, float *strength, float *R)
{
    const keypoint_t *k;
    if (i < 0 || (size_t)i >= kp->num)
        return SIFT3D_FAILURE;
    k = kp->buf + i;
    if (o) *o = k->o;
    if (s) *s = k->s;
    if (xyz_sd) { xyz_sd[0] = k->xd; xyz_sd[1] = k->yd; xyz_sd[2] = k->zd; xyz_sd[3] = k->sd; }
    if (strength) *strength = k->strength;
    if (R) memcpy(R, k->R, sizeof(k->R));
    return SIFT3D_SUCCESS;
}

int sift3d_amd_keypoint_store_set(sift3d_keypoint_store *kp, int n, const int *os,
                                  const double *xyz_sd, const float *strength, const float *R)
{
    int i;
    if (n < 0 || kp_store_resize(kp, (size_t)n))
        return SIFT3D_FAILURE;
    for (i = 0; i < n; i++) {
        keypoint_t *k = kp->buf + i;
        k->o = os[2 * i];
        k->s = os[2 * i + 1];
        k->xd = xyz_sd[4 * i];
        k->yd = xyz_sd[4 * i + 1];
        k->zd = xyz_sd[4 * i + 2];
        k->sd = xyz_sd[4 * i + 3];
        k->strength = strength ? strength[i] : 0.0f;
        memcpy(k->R, R + 9 * i, sizeof(k->R));
    }
    return SIFT3D_SUCCESS;
}

/* ------------------------------------------------------------------------ */
/* icosahedron (init_geometry, sift.c:148-259) -> device face table          */
/* ------------------------------------------------------------------------ */
static int upload_mesh(void)
{
    const float g = golden_ratio;
    const float vert[NVERT][3] = {
        { 0, 1, g }, { 0, -1, g }, { 0, 1, -g }, { 0, -1, -g }, { 1, g, 0 }, { -1, g, 0 },
        { 1, -g, 0 }, { -1, -g, 0 }, { g, 0, 1 }, { -g, 0, 1 }, { g, 0, -1 }, { -g, 0, -1 } };
    static const int faces[NFACES][3] = {
        { 0, 1, 8 }, { 0, 8, 4 }, { 0, 4, 5 }, { 0, 5, 9 }, { 0, 9, 1 }, { 1, 6, 8 },
        { 8, 6, 10 }, { 8, 10, 4 }, { 4, 10, 2 }, { 4, 2, 5 }, { 5, 2, 11 }, { 5, 11, 9 },
        { 9, 11, 7 }, { 9, 7, 1 }, { 1, 7, 6 }, { 3, 6, 7 }, { 3, 7, 11 }, { 3, 11, 2 },
        { 3, 2, 10 }, { 3, 10, 6 } };
    float rec[NFACES * SIFT3D_HIP_FACE_FLOATS];
    int i, j, k, rc;
    const int dev = sift3d_hip_current_device();
    if (dev < 0 || dev >= (int)sizeof(g_mesh_ready))
        return SIFT3D_FAILURE;
    /* (detectors and slab drivers may be created from several threads at once) */
    pthread_mutex_lock(&g_mesh_lock);
    if (g_mesh_ready[dev]) {
        pthread_mutex_unlock(&g_mesh_lock);
        return SIFT3D_SUCCESS;
    }
    for (i = 0; i < NFACES; i++) {
        float v[3][3], a[3], b[3], n[3];
        float *r = rec + i * SIFT3D_HIP_FACE_FLOATS;
        for (j = 0; j < 3; j++) {
            float mag;
            for (k = 0; k < 3; k++)
                v[j][k] = vert[faces[i][j]][k];
            mag = sqrtf(v[j][0] * v[j][0] + v[j][1] * v[j][1] + v[j][2] * v[j][2]);
            /* SIFT3D_CVEC_SCALE(v, 1.0f / mag) expands to x * 1.0f / mag (sift.c:228) */
            for (k = 0; k < 3; k++)
                v[j][k] = v[j][k] * 1.0f / mag;
        }
        for (k = 0; k < 3; k++) {
            a[k] = v[2][k] - v[1][k];
            b[k] = v[1][k] - v[0][k];
        }
        n[0] = a[1] * b[2] - a[2] * b[1];
        n[1] = a[2] * b[0] - a[0] * b[2];
        n[2] = a[0] * b[1] - a[1] * b[0];
        if (n[0] * v[0][0] + n[1] * v[0][1] + n[2] * v[0][2] < 0)
            for (k = 0; k < 3; k++) { /* swap vertices 0,1 -- idx[] is left alone (Q1) */
                const float t = v[0][k];
                v[0][k] = v[1][k];
                v[1][k] = t;
            }
        /* per-face constants of cart2bary (sift.c:276-297) */
        for (k = 0; k < 3; k++) {
            r[0 + k] = v[0][k];
            r[3 + k] = v[1][k] - v[0][k];     /* e1 */
            r[6 + k] = v[2][k] - v[0][k];     /* e2 */
            r[9 + k] = v[0][k] * -1.0f;       /* t  */
        }
        r[12] = r[10] * r[5] - r[11] * r[4];  /* q = t x e1 */
        r[13] = r[11] * r[3] - r[9] * r[5];
        r[14] = r[9] * r[4] - r[10] * r[3];
        r[15] = r[6] * r[12] + r[7] * r[13] + r[8] * r[14]; /* e2 . q */
        for (k = 0; k < 3; k++)
            r[16 + k] = (float)faces[i][k];
    }
    rc = sift3d_hip_set_mesh(rec);
    if (rc == SIFT3D_SUCCESS)
        g_mesh_ready[dev] = 1;
    pthread_mutex_unlock(&g_mesh_lock);
    return rc ? SIFT3D_FAILURE : SIFT3D_SUCCESS;
}

/* ------------------------------------------------------------------------ */
/* detector: parameters, geometry, filter bank                               */
/* ------------------------------------------------------------------------ */
static double level_scale(const sift3d_detector *d, int o, int s)
{
    return d->sigma0 * pow(2.0, o + (double)s / d->num_kp_levels); /* imutil.c:1578-1579 */
}

static void free_filters(sift3d_detector *d)
{
    int i;
    for (i = 0; i < d->nfilt; i++)
        free(d->filt[i].taps);
    free(d->filt);
    d->filt = NULL;
    d->nfilt = 0;
}

/* init_Gauss_filter, imutil.c:1267-1319 */
static int gauss_filter(filter_t *f, double sigma)
{
    const int hw = sigma > 0 ? ((int)ceil(sigma * 3.0) > 1 ? (int)ceil(sigma * 3.0) : 1) : 1;
    const int width = 2 * hw + 1;
    float acc = 0;
    int i;
    f->sigma = sigma;
    f->width = width;
    f->taps = (float *)malloc(sizeof(float) * width);
    if (!f->taps)
        return SIFT3D_FAILURE;
    for (i = 0; i < width; i++) {
        double x = (double)i - hw;
        x /= sigma + DBL_EPSILON;
        f->taps[i] = (float)exp(-0.5 * x * x);
        acc += f->taps[i];
    }
    for (i = 0; i < width; i++)
        f->taps[i] /= acc;
    return SIFT3D_SUCCESS;
}

/* Host filter bank entry for the multi-GPU driver (same code path as the detector's). */
int sift3d_amd_gauss_filter(double sigma, float *taps, int max_taps)
{
    filter_t f;
    int w;
    if (gauss_filter(&f, sigma))
        return -1;
    w = f.width;
    if (w <= max_taps)
        memcpy(taps, f.taps, sizeof(float) * w);
    free(f.taps);
    return w;
}

/* make_gss, imutil.c:1360-1409 */
static int build_filters(sift3d_detector *d)
{
    const int nf = d->ngl;
    int i;
    free_filters(d);
    d->filt = (filter_t *)calloc((size_t)nf, sizeof(filter_t));
    if (!d->filt)
        return SIFT3D_FAILURE;
    d->nfilt = nf;
    for (i = 0; i < nf; i++) {
        const double s_cur = i == 0 ? d->sigma_n : level_scale(d, 0, i - 2);
        const double s_next = level_scale(d, 0, i - 1);
        if (s_cur > s_next) {                      /* imutil.c:1328-1332 */
            ERR("init_Gauss_incremental_filter: s_cur (%f) > s_next (%f) \n", s_cur, s_next);
            return SIFT3D_FAILURE;
        }
        if (gauss_filter(&d->filt[i], sqrt(s_next * s_next - s_cur * s_cur)))
            return SIFT3D_FAILURE;
        /* (any width: filters of more than SIFT3D_HIP_MAX_TAPS taps take the chunked literal kernel,
         * sift3d_hip_fir -- the reference accepts every sigma0 >= 0, sift.c:553-565) */
    }
    return SIFT3D_SUCCESS;
}

static void free_device_pyramid(sift3d_detector *d)
{
    int i;
    if (d->d_g)
        for (i = 0; i < d->num_octaves * d->ngl; i++)
            sift3d_hip_free(d->d_g[i]);
    if (d->d_d)
        for (i = 0; i < d->num_octaves * d->ndl; i++)
            sift3d_hip_free(d->d_d[i]);
    free(d->d_g);
    free(d->d_d);
    d->d_g = d->d_d = NULL;
    sift3d_hip_free(d->d_im);
    sift3d_hip_free(d->d_tmp_a);
    sift3d_hip_free(d->d_tmp_b);
    sift3d_hip_free(d->d_tmp2_a);
    sift3d_hip_free(d->d_tmp2_b);
    sift3d_hip_free(d->d_tmp3_a);
    sift3d_hip_free(d->d_tmp3_b);
    d->d_tmp2_a = d->d_tmp2_b = d->d_tmp3_a = d->d_tmp3_b = NULL;
    sift3d_hip_free(d->d_scalars);
    sift3d_hip_free(d->d_levels);
    sift3d_hip_free(d->d_work);
    sift3d_hip_free(d->d_work2);
    d->d_work2 = NULL;
    d->work2_bytes = 0;
    sift3d_hip_free(d->d_wlut);
    d->d_wlut = NULL;
    sift3d_hip_free(d->d_dpart);
    d->d_dpart = NULL;
    d->dpart_bytes = 0;
    sift3d_hip_free(d->d_otab);
    d->d_otab = NULL;
    d->otab_bytes = 0;
    d->d_im = d->d_tmp_a = d->d_tmp_b = d->d_scalars = NULL;
    d->d_levels = NULL;
    d->d_work = NULL;
    d->work_bytes = 0;
    free(d->h_levels);
    d->h_levels = NULL;
    free(d->odims);
    d->odims = NULL;
    d->num_octaves = 0;
    d->have_pyramid = 0;
}

static void fill_level_table(sift3d_detector *d)
{
    int o, s, k;
    for (o = 0; o < d->num_octaves; o++)
        for (s = 0; s < d->ngl; s++) {
            sift3d_hip_level *L = &d->h_levels[o * d->ngl + s];
            /* octave 0 carries the units of the current image (apply_Sep_FIR_filter
             * copies them from its source, imutil.c:1145); deeper octaves keep what
             * resize_Pyramid gave them (imutil.c:1533,1545-1548) */
            const double *u = o == 0 ? d->units : d->alloc_units;
            double lu[3];
            for (k = 0; k < 3; k++)
                lu[k] = o == 0 ? u[k] : u[k] * ldexp(1.0, o);
            L->data = d->d_g[o * d->ngl + s];
            L->nx = d->odims[o][0];
            L->ny = d->odims[o][1];
            L->nz = d->odims[o][2];
            L->z_off = 0;
            L->nz_glob = d->odims[o][2];
            L->ux = (float)lu[0];
            L->uy = (float)lu[1];
            L->uz = (float)lu[2];
            L->octave = o;
            L->sd = level_scale(d, o, s - 1);
        }
}

/* resize_SIFT3D (sift.c:427-475) + resize_Pyramid (imutil.c:1464-1554) */
static int resize_detector(sift3d_detector *d)
{
    const int ngl = d->num_kp_levels + 3, ndl = d->num_kp_levels + 2;
    int mn, last_octave, o, s, dims[3];
    size_t n0, work = 0;

    free_device_pyramid(d);
    d->ngl = ngl;
    d->ndl = ndl;
    if (!d->have_im)
        return SIFT3D_SUCCESS;

    mn = d->nx < d->ny ? d->nx : d->ny;
    mn = mn < d->nz ? mn : d->nz;
    last_octave = (int)log2((double)mn) - 3;         /* sift.c:442-444 */
    if (last_octave < 0) {
        ERR("resize_SIFT3D: input image is too small: must have at least 8 voxels in each "
            "dimension \n");
        return SIFT3D_FAILURE;
    }
    if (ngl < 2) {                                   /* make_gss, imutil.c:1372-1376 */
        ERR("make_gss: pyr has only %d levels, must have at least 2", ngl);
        return SIFT3D_FAILURE;
    }
    if (level_scale(d, 0, -1) < d->sigma_n) {        /* imutil.c:1582-1588 */
        ERR("set_scales_Pyramid: sigma_n too large for these settings. Max allowed: %f \n",
            level_scale(d, 0, -1) - DBL_EPSILON);
        return SIFT3D_FAILURE;
    }
    d->num_octaves = last_octave + 1;
    memcpy(d->alloc_units, d->units, sizeof(d->units));
    d->odims = (int(*)[3])calloc((size_t)d->num_octaves, sizeof(int[3]));
    d->d_g = (float **)calloc((size_t)d->num_octaves * ngl, sizeof(float *));
    d->d_d = (float **)calloc((size_t)d->num_octaves * ndl, sizeof(float *));
    d->h_levels = (sift3d_hip_level *)calloc((size_t)d->num_octaves * ngl, sizeof(sift3d_hip_level));
    if (!d->odims || !d->d_g || !d->d_d || !d->h_levels)
        return SIFT3D_FAILURE;
    dims[0] = d->nx; dims[1] = d->ny; dims[2] = d->nz;
    for (o = 0; o < d->num_octaves; o++) {
        const size_t n = (size_t)dims[0] * dims[1] * dims[2];
        const size_t w = sift3d_hip_extrema_work_bytes(dims[0], dims[1], dims[2], ndl - 2);
        memcpy(d->odims[o], dims, sizeof(dims));
        for (s = 0; s < ngl; s++)
            if (!(d->d_g[o * ngl + s] = (float *)sift3d_hip_malloc(n * sizeof(float))))
                return SIFT3D_FAILURE;
        /* (DoG levels are allocated only where an octave needs them stored: ensure_dog_octave) */
        work = w > work ? w : work;
        if (o >= 1 && o < 64) {
            d->work2_off[o] = d->work2_bytes;
            d->work2_bytes += (w + 255) & ~(size_t)255;
        }
        for (s = 0; s < 3; s++)
            dims[s] /= 2;                            /* imutil.c:1545-1547 */
    }
    n0 = (size_t)d->nx * d->ny * d->nz;
    d->d_im = (float *)sift3d_hip_malloc(n0 * sizeof(float));
    d->d_tmp_a = (float *)sift3d_hip_malloc(n0 * sizeof(float));
    d->d_tmp_b = (float *)sift3d_hip_malloc(n0 * sizeof(float));
    if (d->num_octaves > 1) {
        const size_t n1 = (size_t)d->odims[1][0] * d->odims[1][1] * d->odims[1][2];
        d->d_tmp2_a = (float *)sift3d_hip_malloc(n1 * sizeof(float));
        d->d_tmp2_b = (float *)sift3d_hip_malloc(n1 * sizeof(float));
        d->d_tmp3_a = (float *)sift3d_hip_malloc(n1 * sizeof(float));
        d->d_tmp3_b = (float *)sift3d_hip_malloc(n1 * sizeof(float));
        if (!d->d_tmp2_a || !d->d_tmp2_b || !d->d_tmp3_a || !d->d_tmp3_b)
            return SIFT3D_FAILURE;
    }
    d->d_scalars = (float *)sift3d_hip_malloc(sizeof(float) * (8 + 2 * (size_t)d->num_octaves * ndl));
    d->d_levels = (sift3d_hip_level *)sift3d_hip_malloc(sizeof(sift3d_hip_level) *
                                                        (size_t)d->num_octaves * ngl);
    d->d_work = sift3d_hip_malloc(work);
    d->work_bytes = work;
    if (d->work2_bytes && !(d->d_work2 = sift3d_hip_malloc(d->work2_bytes)))
        return SIFT3D_FAILURE;
    d->d_wlut = (float *)sift3d_hip_malloc(sizeof(float) *
                                           sift3d_hip_describe_wlut_floats(d->num_octaves * ngl));
    if (!d->d_wlut || !d->d_im || !d->d_tmp_a || !d->d_tmp_b || !d->d_scalars || !d->d_levels || !d->d_work)
        return SIFT3D_FAILURE;
    return build_filters(d);
}

int sift3d_detector_set_peak_thresh(sift3d_detector *const d, const double v)
{
    if (v <= 0.0 || v > 1) {                         /* sift.c:501-505 */
        ERR("sift3d_detector peak_thresh must be in the interval (0, 1]. Provided: %f \n", v);
        return SIFT3D_FAILURE;
    }
    d->peak_thresh = v;
    return SIFT3D_SUCCESS;
}

int sift3d_amd_detector_set_cuboid_extrema(sift3d_detector *d, int on)
{
    if (!d)
        return SIFT3D_FAILURE;
    d->cuboid_extrema = on ? 1 : 0;
    return SIFT3D_SUCCESS;
}

int sift3d_amd_detector_set_dogmax_pass(sift3d_detector *d, int on)
{
    if (!d)
        return SIFT3D_FAILURE;
    d->est0 = on ? 0 : 1;
    return SIFT3D_SUCCESS;
}

int sift3d_amd_detector_set_serial_orientation(sift3d_detector *d, int on)
{
    if (!d)
        return SIFT3D_FAILURE;
    d->orient_serial = on != 0;
    return SIFT3D_SUCCESS;
}

int sift3d_amd_detector_set_exact_descriptors(sift3d_detector *d, int mode)
{
    if (!d || mode < -1 || mode > 1)
        return SIFT3D_FAILURE;
    d->exact_desc = mode;
    return SIFT3D_SUCCESS;
}

/* First level-in-octave index (s + 1, Gaussian index) whose keypoints take the reference-order descriptor
 * kernel (sift3d_hip_describe_ex); ngl when none does.  A window of level s holds ~ (2 * half)^3 /
 * (ux uy uz) voxels with half = 2 * sd * 7.071 / sqrt(2) (sift.c:1453-1456) and sd / units the same in
 * every octave (Q9); a bin receives 1/32 of them.  Measured: 5e-6 maximal relative difference to the
 * reference over 3.3e7 bins at 1.3e5 window voxels (512^3 fixture, fast commit); the difference grows
 * with the square root of the terms per bin: 1.9e5 voxels keep it at ~6e-6, inside the 1e-5 bar. */
static int exact_desc_first_level(int mode, int ngl, int K, double sigma0, const double *units)
{
    int lv;
    if (mode > 0)
        return 0;
    if (mode < 0)
        return ngl;
    for (lv = 0; lv < ngl; lv++) {
        const double sd = sigma0 * pow(2.0, (double)(lv - 1) / K);
        const double side = 2.0 * (2.0 * sd * 7.071067812 / 1.4142135623730951);
        if (side * side * side / (units[0] * units[1] * units[2]) > 1.9e5)
            return lv;
    }
    return ngl;
}

int sift3d_detector_set_corner_thresh(sift3d_detector *const d, const double v)
{
    if (v < 0.0 || v > 1.0) {                        /* sift.c:515-519 */
        ERR("sift3d_detector corner_thresh must be in the interval [0, 1]. Provided: %f \n", v);
        return SIFT3D_FAILURE;
    }
    d->corner_thresh = v;
    return SIFT3D_SUCCESS;
}

int sift3d_detector_set_num_kp_levels(sift3d_detector *const d, const unsigned int v)
{
    d->num_kp_levels = (int)v;                       /* sift.c:527-533 */
    return resize_detector(d);
}

/* set_scales_SIFT3D, sift.c:478-496 */
static int set_scales(sift3d_detector *d, double sigma0, double sigma_n)
{
    const double old0 = d->sigma0, oldn = d->sigma_n;
    d->sigma0 = sigma0;
    d->sigma_n = sigma_n;
    if (!d->num_octaves)
        return SIFT3D_SUCCESS;
    if (level_scale(d, 0, -1) < sigma_n) {           /* imutil.c:1582-1588 */
        ERR("set_scales_Pyramid: sigma_n too large for these settings. Max allowed: %f \n",
            level_scale(d, 0, -1) - DBL_EPSILON);
        d->sigma0 = old0;
        d->sigma_n = oldn;
        return SIFT3D_FAILURE;
    }
    d->have_pyramid = 0;
    return build_filters(d);
}

int sift3d_detector_set_sigma_n(sift3d_detector *const d, const double v)
{
    if (v < 0.0) {                                   /* sift.c:542-546 */
        ERR("sift3d_detector sigma_n must be nonnegative. Provided: %f \n", v);
        return SIFT3D_FAILURE;
    }
    return set_scales(d, d->sigma0, v);
}

int sift3d_detector_set_sigma0(sift3d_detector *const d, const double v)
{
    if (v < 0.0) {                                   /* sift.c:558-562 */
        ERR("sift3d_detector sigma0 must be nonnegative. Provided: %f \n", v);
        return SIFT3D_FAILURE;
    }
    return set_scales(d, v, d->sigma_n);
}

sift3d_detector *sift3d_make_detector()
{
    sift3d_detector *d = (sift3d_detector *)calloc(1, sizeof(*d));
    if (!d)
        return NULL;
    d->peak_thresh = peak_thresh_default;
    d->est0 = 1;
    d->corner_thresh = corner_thresh_default;
    d->sigma_n = sigma_n_default;
    d->sigma0 = sigma0_default;
    d->num_kp_levels = num_kp_levels_default;
    d->ngl = d->num_kp_levels + 3;
    d->ndl = d->num_kp_levels + 2;
    return d;
}

void sift3d_free_detector(sift3d_detector *d)
{
    int i;
    if (!d)
        return;
    if (d->stream)
        sift3d_hip_stream_sync(d->stream);
    free_device_pyramid(d);
    free_filters(d);
    sift3d_hip_free(d->d_in);
    sift3d_hip_free(d->d_cand);
    sift3d_hip_host_free(d->h_cand);
    sift3d_hip_host_free(d->h_R);
    sift3d_hip_host_free(d->h_keep);
    sift3d_hip_host_free(d->h_kp);
    for (i = 0; i < 8; i++) {
        sift3d_hip_event_destroy(d->ev[i]);
        sift3d_hip_event_destroy(d->ev_chunk[i]);
    }
    sift3d_hip_event_destroy(d->ev_fork);
    sift3d_hip_event_destroy(d->ev_join);
    sift3d_hip_event_destroy(d->ev_join2);
    sift3d_hip_event_destroy(d->ev_part);
    for (i = 0; i < SIFT3D_AMD_TIMED_BLURS * 3; i++)
        sift3d_hip_event_destroy(d->ev_blur[i / 3][i % 3]);
    sift3d_hip_event_destroy(d->ev_pyr[0]);
    sift3d_hip_event_destroy(d->ev_pyr[1]);
    for (i = 0; i < 32; i++)
        sift3d_hip_event_destroy(d->ev_oct[i]);
    sift3d_hip_stream_destroy(d->side_stream);
    sift3d_hip_stream_destroy(d->oct_stream);
    sift3d_hip_stream_destroy(d->copy_stream);
    sift3d_hip_stream_destroy(d->stream);
    free(d);
}

/* seconds between two stage events; NaN when the pair is not a completed pair of one call (a failed
 * query must not read as a duration) */
static double stage_seconds(void *a, void *b)
{
    const double ms = sift3d_hip_event_elapsed_ms(a, b);
    return ms >= 0.0 ? 1e-3 * ms : (double)NAN;
}

/* stage seconds of the last calls; the device-side ones are read from the stage events here, not on the
 * path of a step (both calls end with a stream synchronisation, so the events are complete) */
const double *sift3d_amd_timings(const sift3d_detector *dc)
{
    sift3d_detector *d = (sift3d_detector *)dc;
    if (d->t_pending & 1) {
        d->t[0] = stage_seconds(d->ev[0], d->ev[1]);
        int b, last = -1;
        /* the pyramid ends with the LAST of its chains (octave 0 on the main stream; the first levels of
         * the smaller octaves; their last levels); the stages after it start on the main stream when
         * octave 0 is complete, so [2] and [3] overlap the tail of [1] */
        d->t[1] = stage_seconds(d->ev[1], d->ev[2]);
        if (d->pyr_chains)
            for (b = 0; b < 2; b++) {
                const double tc = stage_seconds(d->ev[1], d->ev_pyr[b]);
                if (tc > d->t[1] || isnan(tc))
                    d->t[1] = tc;
            }
        d->t[2] = stage_seconds(d->ev[2], d->ev[3]);
        d->t[3] = stage_seconds(d->ev[3], d->ev[4]);
        d->t[4] = stage_seconds(d->ev[4], d->ev[5]);
        d->t[6] = d->t[1];
        for (b = 0; b < SIFT3D_AMD_TIMED_BLURS; b++) {
            const int on = (d->yz_timed >> b) & 1;
            d->t[10 + b] = on ? stage_seconds(d->ev_blur[b][0], d->ev_blur[b][1]) : 0.0;
            d->t[10 + SIFT3D_AMD_TIMED_BLURS + b] = on ? stage_seconds(d->ev_blur[b][1], d->ev_blur[b][2]) : 0.0;
            if (on)
                last = b;
        }
        d->t[9] = last >= 0 ? d->t[10 + SIFT3D_AMD_TIMED_BLURS + last] : 0.0;
        d->t[10 + 2 * SIFT3D_AMD_TIMED_BLURS] = stage_seconds(d->ev[0], d->ev[5]);
        d->t[10 + 2 * SIFT3D_AMD_TIMED_BLURS + 2] = d->parts_timed ? stage_seconds(d->ev[0], d->ev_part) : 0.0;
        d->t[10 + 2 * SIFT3D_AMD_TIMED_BLURS + 3] = d->parts_timed ? stage_seconds(d->ev[0], d->ev_join) : 0.0;
    }
    if (d->t_pending & 2)
        d->t[5] = stage_seconds(d->ev[6], d->ev[7]);
    d->t_pending = 0;
    return d->t;
}
int sift3d_amd_num_candidates(const sift3d_detector *d) { return d->ncand; }

int sift3d_amd_describe_clock(const sift3d_detector *d, double *cycles, double *seconds)
{
    uint64_t c = 0, t = 0;
    if (!d || !d->d_wlut || !d->num_octaves || !cycles || !seconds ||
        sift3d_hip_describe_clock(d->d_wlut, d->num_octaves * d->ngl, 0, &c, &t, d->stream))
        return SIFT3D_FAILURE;
    *cycles = (double)c;
    *seconds = (double)t * 1e-8;
    return SIFT3D_SUCCESS;
}

/* max|DoG| of every level of the last detect call (the dogmax scan, sift.c:821-826): out[o * ndl + s],
 * capacity `cap` floats; returns the number of values or -1 */
int sift3d_amd_detector_dogmax(const sift3d_detector *d, float *out, int cap)
{
    const int n = d ? d->num_octaves * d->ndl : 0;
    if (!d || !out || !d->have_pyramid || !d->d_scalars || n < 1 || cap < n)
        return -1;
    if (sift3d_hip_memcpy_d2h(out, d->d_scalars + 8, sizeof(float) * (size_t)n, d->stream) ||
        sift3d_hip_stream_sync(d->stream))
        return -1;
    return n;
}

/* ------------------------------------------------------------------------ */
/* tracing: roctx ranges around the stages (rocprofv3 --marker-trace) */
/* ------------------------------------------------------------------------ */
/* The marker library is loaded at run time, at the first call; without it the ranges cost a pointer
 * test.  Ranges are started / stopped by id, so a call that fails half-way leaves no unbalanced
 * stack behind. */
static struct {
    pthread_once_t once;
    uint64_t (*start)(const char *);
    void (*stop)(uint64_t);
} g_roctx = { PTHREAD_ONCE_INIT, NULL, NULL };

static void roctx_load(void)
{
    /* (rocprofv3's marker library first, the roctracer one as a fallback; SIFT3D_AMD_ROCTX=0 disables) */
    static const char *names[] = { "librocprofiler-sdk-roctx.so.1", "librocprofiler-sdk-roctx.so",
                                   "libroctx64.so.4", "libroctx64.so" };
    void *lib = NULL;
    size_t i;
    const char *e = getenv("SIFT3D_AMD_ROCTX");
    if (e && e[0] == '0')
        return;
    for (i = 0; i < sizeof(names) / sizeof(names[0]) && !lib; i++)
        lib = dlopen(names[i], RTLD_NOW | RTLD_GLOBAL);
    if (!lib)
        return;
    *(void **)(&g_roctx.start) = dlsym(lib, "roctxRangeStartA");
    *(void **)(&g_roctx.stop) = dlsym(lib, "roctxRangeStop");
    if (!g_roctx.start || !g_roctx.stop)
        g_roctx.start = NULL;
}

static uint64_t range_start(const char *name)
{
    pthread_once(&g_roctx.once, roctx_load);
    return g_roctx.start ? g_roctx.start(name) : 0;
}

static void range_stop(uint64_t id)
{
    if (g_roctx.start)
        g_roctx.stop(id);
}

/* ------------------------------------------------------------------------ */
/* the hot path                                                              */
/* ------------------------------------------------------------------------ */
static int ensure_device(sift3d_detector *d)
{
    int i;
    if (d->stream) {
        /* a detector lives on the device it first ran on (its streams, pyramids, tables) */
        if (sift3d_hip_current_device() != d->device) {
            ERR("sift3d_amd: this detector belongs to HIP device %d, the current device is %d \n",
                d->device, sift3d_hip_current_device());
            return SIFT3D_FAILURE;
        }
        return SIFT3D_SUCCESS;
    }
    if (!sift3d_amd_device_available()) {
        ERR("sift3d_amd: no HIP device is available; this library has no CPU path \n");
        return SIFT3D_FAILURE;
    }
    d->device = sift3d_hip_current_device();
    if (!(d->stream = sift3d_hip_stream_create()) || !(d->copy_stream = sift3d_hip_stream_create()) ||
        !(d->oct_stream = sift3d_hip_stream_create_high()) ||
        !(d->side_stream = sift3d_hip_stream_create_high()) || !(d->ev_fork = sift3d_hip_event_create()) ||
        !(d->ev_join = sift3d_hip_event_create()) || !(d->ev_join2 = sift3d_hip_event_create()) ||
        !(d->ev_part = sift3d_hip_event_create()) ||
        !(d->ev_pyr[0] = sift3d_hip_event_create()) || !(d->ev_pyr[1] = sift3d_hip_event_create()))
        return SIFT3D_FAILURE;
    for (i = 0; i < 32; i++)
        if (!(d->ev_oct[i] = sift3d_hip_event_create()))
            return SIFT3D_FAILURE;
    for (i = 0; i < 8; i++)
        if (!(d->ev[i] = sift3d_hip_event_create()) || !(d->ev_chunk[i] = sift3d_hip_event_create()))
            return SIFT3D_FAILURE;
    for (i = 0; i < SIFT3D_AMD_TIMED_BLURS * 3; i++)
        if (!(d->ev_blur[i / 3][i % 3] = sift3d_hip_event_create()))
            return SIFT3D_FAILURE;
    return upload_mesh();
}

/* One-time device-side tables for callers of the stage ABI that do not go through a
 * detector (the multi-GPU driver). */
int sift3d_amd_init(void)
{
    if (!sift3d_amd_device_available()) {
        ERR("sift3d_amd: no HIP device is available; this library has no CPU path \n");
        return SIFT3D_FAILURE;
    }
    return upload_mesh();
}

static int ensure_cand_capacity(sift3d_detector *d, uint32_t cap)
{
    if (cap <= d->cand_cap)
        return SIFT3D_SUCCESS;
    sift3d_hip_free(d->d_cand);
    sift3d_hip_host_free(d->h_cand);
    sift3d_hip_host_free(d->h_R);
    sift3d_hip_host_free(d->h_keep);
    d->cand_cap = 0;
    d->d_cand = (sift3d_hip_cand *)sift3d_hip_malloc(sizeof(sift3d_hip_cand) * (size_t)cap);
    d->h_cand = (sift3d_hip_cand *)sift3d_hip_host_alloc(sizeof(sift3d_hip_cand) * (size_t)cap);
    d->h_R = (float *)sift3d_hip_host_alloc(sizeof(float) * 9 * (size_t)cap);
    /* (+ 4 page-locked words behind the flags: the candidate counts land there -- a copy into pageable memory
     * would hold the host until it has been carried out) */
    d->h_keep = (int32_t *)sift3d_hip_host_alloc(sizeof(int32_t) * ((size_t)cap + 4));
    if (!d->d_cand || !d->h_cand || !d->h_R || !d->h_keep)
        return SIFT3D_FAILURE;
    d->cand_cap = cap;
    return SIFT3D_SUCCESS;
}

/* DoG levels of one octave in memory -- only for configurations the DoG-free extrema sweep
 * does not cover (sift3d_hip_extrema_gauss6) */
static int ensure_dog_octave(sift3d_detector *d, int o)
{
    const size_t n = (size_t)d->odims[o][0] * d->odims[o][1] * d->odims[o][2];
    int s;
    for (s = 0; s < d->ndl; s++)
        if (!d->d_d[o * d->ndl + s] &&
            !(d->d_d[o * d->ndl + s] = (float *)sift3d_hip_malloc(n * sizeof(float))))
            return SIFT3D_FAILURE;
    return SIFT3D_SUCCESS;
}

/* apply_Sep_FIR_filter (imutil.c:1127-1206) on the device: x, y, z passes, the two
 * intermediates in scratch volumes, no permute copies */
/* d_scale_max (first blur of the pyramid only, else NULL): the blur of src / *d_scale_max -- im_scale folded
 * into the x pass; returns 2 without doing anything when the configuration's x pass cannot do that (the
 * caller then scales the image first) */
static int blur_level(sift3d_detector *d, const float *src, float *dst, const int *dims,
                      const double *lu, const filter_t *f, void *stream, float *tmp_a, float *tmp_b,
                      int slot, const float *d_scale_max)
{
    const float *in = src;
    float *outs[3];
    int ax;
    outs[0] = tmp_a;
    outs[1] = tmp_b;
    outs[2] = dst;
    /* tap spacing 1 on y and z (octave 0 of a unit-spaced volume): x pass, then the fused
     * y+z kernel -- the y-pass result never goes to HBM */
    if ((float)(1.0 / lu[1]) == 1.0f && (float)(1.0 / lu[2]) == 1.0f) {
        sift3d_hip_fir_args a;
        int rc;
        memset(&a, 0, sizeof(a));
        a.src = src; a.dst = tmp_a;
        a.nx = dims[0]; a.ny = dims[1]; a.nz = dims[2];
        a.axis = 0; a.width = f->width; a.taps = f->taps;
        a.unit_factor = (float)(1.0 / lu[0]);
        a.n_glob = dims[2]; a.z_lo = 0; a.z_hi = dims[2];
        if (slot >= SIFT3D_AMD_TIMED_BLURS)
            slot = -1;
        if (slot >= 0)
            sift3d_hip_event_record(d->ev_blur[slot][0], stream);
        if (d_scale_max) {
            rc = sift3d_hip_fir_x_scaled(&a, d_scale_max, stream);
            if (rc == 1)
                return 2;
            if (rc != SIFT3D_SUCCESS)
                return SIFT3D_FAILURE;
        } else if (sift3d_hip_fir(&a, stream))
            return SIFT3D_FAILURE;
        if (slot >= 0)
            sift3d_hip_event_record(d->ev_blur[slot][1], stream);
        rc = sift3d_hip_fir_yz_u1(tmp_a, dst, dims[0], dims[1], dims[2], f->taps, f->width,
                                  dims[2], 0, 0, dims[2], stream);
        if (rc == SIFT3D_SUCCESS) {
            if (slot >= 0) {
                sift3d_hip_event_record(d->ev_blur[slot][2], stream);
                d->yz_timed |= 1u << slot;
            }
            return SIFT3D_SUCCESS;
        }
        if (rc != 1)
            return SIFT3D_FAILURE;
        in = tmp_a;                 /* not covered: finish with separate y and z passes */
        for (ax = 1; ax < 3; ax++) {
            memset(&a, 0, sizeof(a));
            a.src = in; a.dst = outs[ax];
            a.nx = dims[0]; a.ny = dims[1]; a.nz = dims[2];
            a.axis = ax; a.width = f->width; a.taps = f->taps;
            a.unit_factor = (float)(1.0 / lu[ax]);
            a.n_glob = dims[2]; a.z_lo = 0; a.z_hi = dims[2];
            if (sift3d_hip_fir(&a, stream))
                return SIFT3D_FAILURE;
            in = outs[ax];
        }
        return SIFT3D_SUCCESS;
    }
    if (d_scale_max)
        return 2;
    for (ax = 0; ax < 3; ax++) {
        sift3d_hip_fir_args a;
        memset(&a, 0, sizeof(a));
        a.src = in;
        a.dst = outs[ax];
        a.nx = dims[0]; a.ny = dims[1]; a.nz = dims[2];
        a.axis = ax;
        a.width = f->width;
        a.taps = f->taps;
        a.unit_factor = (float)(1.0 / lu[ax]);     /* unit = 1.0: sift.c:675, imutil.c:754-755 */
        a.n_glob = dims[2];
        a.off = 0;
        a.z_lo = 0;
        a.z_hi = dims[2];
        if (sift3d_hip_fir(&a, stream))
            return SIFT3D_FAILURE;
        in = outs[ax];
    }
    return SIFT3D_SUCCESS;
}

static void level_units(const sift3d_detector *d, int o, double *lu)
{
    int k;
    for (k = 0; k < 3; k++)
        lu[k] = o == 0 ? d->units[k] : d->alloc_units[k] * ldexp(1.0, o);
}

/* device part of sift3d_detect_keypoints (sift.c:1217-1249) */
/* Host loops over the candidate / keypoint lists (10^5 records at 512^3) run between the last kernel of
 * one stage and the first of the next, with the device idle: a few threads, statically split so that the
 * order of the records -- the reference's scan order -- is kept. */
/* octaves whose dogmax scan is gathered by the extrema sweep: those large enough for the saved bytes to
 * outweigh four more (short) launches */
#define EST_OCTAVE(d, o) ((d)->est0 && (size_t)(d)->odims[o][0] * (d)->odims[o][1] * (d)->odims[o][2] >= ((size_t)1 << 21))
#define HOST_THREADS_MAX 8
static int host_threads(size_t n)
{
    int t = omp_get_num_procs();
    if (t > HOST_THREADS_MAX)
        t = HOST_THREADS_MAX;
    if (n < 4096 || t < 1)
        t = 1;
    return t;
}

/* scratch of the orientation kernels: sized by the level count and the candidate capacity */
static int orient_scratch(sift3d_detector *d)
{
    const size_t need = sift3d_hip_orient_tab_bytes(d->num_octaves * d->ngl, d->cand_cap);
    if (need > d->otab_bytes) {
        sift3d_hip_free(d->d_otab);
        d->otab_bytes = 0;
        d->d_otab = sift3d_hip_malloc(need);
        /* zeroed once: the tables carry a validity mark (they are kept between calls); complete before any
         * stream's kernels use it */
        if (!d->d_otab || sift3d_hip_memset(d->d_otab, 0, need, d->stream) || sift3d_hip_stream_sync(d->stream))
            return SIFT3D_FAILURE;
        d->otab_bytes = need;
    }
    return SIFT3D_SUCCESS;
}

/* Candidates first .. first + n - 1 (all of levels lv_lo .. lv_hi - 1) through the orientation kernels.  R and
 * the keep flags are written by the kernels straight into the page-locked host arrays (mapped into the device's
 * address space; only kept candidates' matrices are written): no device staging, no copy after the kernels. */
static int orient_part(sift3d_detector *d, int lv_lo, int lv_hi, uint32_t first, uint32_t n, int slot, void *stream)
{
    float *r_view = (float *)sift3d_hip_host_device_ptr(d->h_R);
    int32_t *k_view = (int32_t *)sift3d_hip_host_device_ptr(d->h_keep);
    if (!r_view || !k_view)
        return SIFT3D_FAILURE;
    return sift3d_hip_orient_tab_part(d->d_levels, d->num_octaves * d->ngl, lv_lo, lv_hi, d->d_cand, first, n,
                                      d->corner_thresh, r_view, k_view, d->orient_serial ? NULL : d->d_otab,
                                      d->cand_cap, slot, stream);
}

static int detect_on_device(sift3d_detector *d, const float *d_vol, int nx, int ny, int nz,
                            double ux, double uy, double uz, sift3d_keypoint_store *kp, int pyramid_only)
{
    const size_t n0 = (size_t)nx * ny * nz;
    const int dims_changed = !d->have_im || d->nx != nx || d->ny != ny || d->nz != nz ||
                             !d->num_octaves;
    const double t_start = now_s();
    uint32_t count = 0;
    uint64_t rng;
    int o, s, attempt, side, overlap = 0, im_stored = 0, oriented = 0;

    /* set_im_SIFT3D, sift.c:629-659 */
    d->have_im = 1;
    d->nx = nx; d->ny = ny; d->nz = nz;
    d->units[0] = ux; d->units[1] = uy; d->units[2] = uz;
    d->have_pyramid = 0;
    d->im_valid = 0;                    /* (both are set again only by a call that succeeds) */
    d->last_vol = NULL;
    d->t_pending &= ~1;                 /* the stage events are re-recorded from here on */
    if (dims_changed && resize_detector(d)) {
        d->have_im = 0;
        return SIFT3D_FAILURE;
    }
    fill_level_table(d);
    if (sift3d_hip_memcpy_h2d(d->d_levels, d->h_levels,
                              sizeof(sift3d_hip_level) * (size_t)d->num_octaves * d->ngl, d->stream))
        return SIFT3D_FAILURE;

    rng = range_start("sift3d: max|v|");
    sift3d_hip_event_record(d->ev[0], d->stream);
    if (sift3d_hip_memset(d->d_scalars, 0, sizeof(float) * (8 + 2 * (size_t)d->num_octaves * d->ndl),
                          d->stream) ||
        sift3d_hip_absmax(d_vol, n0, d->d_scalars, d->stream))
        return SIFT3D_FAILURE;

    /* build_gpyr, sift.c:662-711.  The first blur reads the volume itself and divides every sample by the
     * maximum as it stages it (im_scale, imutil.c:698-713): the scaled image -- read by nothing else -- is
     * not stored (8 B/voxel less; sift3d_amd_copy_level forms it on demand from `last_vol`).  Where the x
     * pass cannot do that (other tap spacings) the image is scaled first, as before. */
#ifdef SIFT3D_AMD_DIAG
    /* diagnostic build only (profiles/): 1 = octave 0's last blurs are not held back, 2 = the stages after the
     * pyramid wait for all of its chains (the schedule of round 4) */
    const int sched = getenv("SIFT3D_AMD_SCHED") ? atoi(getenv("SIFT3D_AMD_SCHED")) : 0;
#else
    const int sched = 0;
#endif
    d->yz_timed = 0;
    d->parts_timed = 0;
    d->pyr_chains = 0;
    /* Default configuration on every octave (and a second stream at hand): the stages after the pyramid run
     * octave 0 on the main stream and the short launches of octaves >= 1 beside it -- in the DoG stage and
     * again for the extrema sweeps, whose results are then emitted in octave order. */
    side = !d->cuboid_extrema && d->ngl == 6 && d->num_octaves > 1 && d->num_octaves <= 32 && d->d_work2;
    for (o = 0; o < d->num_octaves && side; o++)
        side = (d->odims[o][0] & 3) == 0 && d->odims[o][2] >= 3;
    range_stop(rng);
    rng = range_start("sift3d: Gaussian pyramid");
    sift3d_hip_event_record(d->ev[1], d->stream);
    {
        double lu[3];
        int rc;
        level_units(d, 0, lu);
        rc = blur_level(d, d_vol, d->d_g[0], d->odims[0], lu, &d->filt[0], d->stream, d->d_tmp_a,
                        d->d_tmp_b, 0, d->d_scalars);
        if (rc == 2) {
            if (sift3d_hip_scale(d_vol, d->d_im, n0, d->d_scalars, d->stream) ||
                blur_level(d, d->d_im, d->d_g[0], d->odims[0], lu, &d->filt[0], d->stream, d->d_tmp_a,
                           d->d_tmp_b, 0, NULL))
                return SIFT3D_FAILURE;
            im_stored = 1;
        } else if (rc != SIFT3D_SUCCESS) {
            return SIFT3D_FAILURE;
        }
    }
    {
        /* Octave o + 1 starts from level max(s_end - 2, first_level) of octave o (sift.c:696-704);
         * the levels after it belong to octave o alone.  So once that level of octave 0 exists, the
         * smaller octaves -- short kernels that cannot fill the device -- are built on a second
         * stream BESIDE the last levels of octave 0, their own last levels (which nothing waits
         * for) on a third, and all are joined before the DoG stage. */
        const int s_end = d->ngl - 2;
        const int ds = s_end - 2 > -1 ? s_end - 2 : -1;
        const int forked = d->num_octaves > 1 && d->num_octaves <= 32 && ds + 1 < d->ngl - 1;
        int o0_rest = 0;         /* first level of octave 0 that is still to be enqueued (0: none) */
        for (o = 0; o < d->num_octaves; o++) {
            double lu[3];
            void *st = (o > 0 && forked) ? d->oct_stream : d->stream;
            float *ta = (o > 0 && forked) ? d->d_tmp2_a : d->d_tmp_a;
            float *tb = (o > 0 && forked) ? d->d_tmp2_b : d->d_tmp_b;
            level_units(d, o, lu);
            if (o == 0 && forked) {
                /* octave 0 up to the source level, then the fork */
                for (s = 1; s <= ds + 1; s++)
                    if (blur_level(d, d->d_g[s - 1], d->d_g[s], d->odims[0], lu, &d->filt[s], d->stream,
                                   d->d_tmp_a, d->d_tmp_b, s, NULL))
                        return SIFT3D_FAILURE;
                if (sift3d_hip_event_record(d->ev_fork, d->stream) ||
                    sift3d_hip_stream_wait_event(d->oct_stream, d->ev_fork))
                    return SIFT3D_FAILURE;
                if (sift3d_hip_downsample2(d->d_g[ds + 1], d->odims[0][0], d->odims[0][1],
                                           d->d_g[d->ngl], d->odims[1][0], d->odims[1][1], d->odims[1][2],
                                           d->oct_stream))
                    return SIFT3D_FAILURE;
                /* The last levels of octave 0 are held back until octave 1 has reached ITS source level.  The
                 * fused y+z kernel keeps one or two workgroups on every CU for its whole duration: beside it
                 * the passes of octave 1 -- which the whole chain of smaller octaves waits for -- get what
                 * registers and LDS it leaves (in the step 2-7x their stand-alone time), and it loses
                 * bandwidth to them.  Octave 1's first levels alone take 0.4 ms; the pyramid's total does
                 * not change (3.47-3.51 against 3.48-3.49 ms), the two large kernels' interference does. */
                o0_rest = s;     /* (forked: octave 1 exists and reaches the point where they are enqueued) */
                if (sched & 1) {
                    for (; o0_rest < d->ngl; o0_rest++)
                        if (blur_level(d, d->d_g[o0_rest - 1], d->d_g[o0_rest], d->odims[0], lu, &d->filt[o0_rest],
                                       d->stream, d->d_tmp_a, d->d_tmp_b, o0_rest, NULL))
                            return SIFT3D_FAILURE;
                    o0_rest = 0;
                }
                continue;
            }
            for (s = 1; s < d->ngl; s++) {
                if (forked && s == ds + 2) {
                    /* the rest of this octave leaves the critical chain */
                    if (sift3d_hip_event_record(d->ev_oct[o], st) ||
                        sift3d_hip_stream_wait_event(d->side_stream, d->ev_oct[o]))
                        return SIFT3D_FAILURE;
                    if (o != d->num_octaves - 1 &&
                        sift3d_hip_downsample2(d->d_g[o * d->ngl + ds + 1], d->odims[o][0], d->odims[o][1],
                                               d->d_g[(o + 1) * d->ngl], d->odims[o + 1][0],
                                               d->odims[o + 1][1], d->odims[o + 1][2], st))
                        return SIFT3D_FAILURE;
                    if (o0_rest) {
                        /* ... and now octave 0's last levels, on the main stream */
                        double lu0[3];
                        int s0i;
                        level_units(d, 0, lu0);
                        if (sift3d_hip_stream_wait_event(d->stream, d->ev_oct[o]))
                            return SIFT3D_FAILURE;
                        for (s0i = o0_rest; s0i < d->ngl; s0i++)
                            if (blur_level(d, d->d_g[s0i - 1], d->d_g[s0i], d->odims[0], lu0, &d->filt[s0i],
                                           d->stream, d->d_tmp_a, d->d_tmp_b, s0i, NULL))
                                return SIFT3D_FAILURE;
                        o0_rest = 0;
                    }
                    st = d->side_stream;
                    ta = d->d_tmp3_a;
                    tb = d->d_tmp3_b;
                }
                if (blur_level(d, d->d_g[o * d->ngl + s - 1], d->d_g[o * d->ngl + s], d->odims[o], lu,
                               &d->filt[s], st, ta, tb, o == 0 ? s : -1, NULL)) /* gauss_octave[s], sift.c:689 */
                    return SIFT3D_FAILURE;
            }
            if (o != d->num_octaves - 1 && !forked) {
                if (sift3d_hip_downsample2(d->d_g[o * d->ngl + ds + 1], d->odims[o][0], d->odims[o][1],
                                           d->d_g[(o + 1) * d->ngl], d->odims[o + 1][0],
                                           d->odims[o + 1][1], d->odims[o + 1][2], st))
                    return SIFT3D_FAILURE;
            }
        }
        if (forked) {
            /* Round 5: the main stream does NOT wait for the chains of the smaller octaves here.  Octave 0's
             * DoG maxima and extrema sweep (0.9 ms, device-filling) need octave 0's levels only and start
             * when its last blur ends; the two side chains wait for EACH OTHER (an octave's levels 4, 5 are
             * built on the side stream, its first ones on the octave stream) and go on with their own
             * octaves' DoG maxima and sweeps.  All three meet again before scan + emission. */
            overlap = side && !pyramid_only && !(sched & 2);
            d->pyr_chains = 1;
            if (sift3d_hip_event_record(d->ev_pyr[0], d->oct_stream) ||
                sift3d_hip_event_record(d->ev_pyr[1], d->side_stream))
                return SIFT3D_FAILURE;
            if (overlap) {
                if (sift3d_hip_stream_wait_event(d->oct_stream, d->ev_pyr[1]) ||
                    sift3d_hip_stream_wait_event(d->side_stream, d->ev_pyr[0]))
                    return SIFT3D_FAILURE;
            } else if (sift3d_hip_stream_wait_event(d->stream, d->ev_pyr[0]) ||
                       sift3d_hip_stream_wait_event(d->stream, d->ev_pyr[1])) {
                return SIFT3D_FAILURE;
            }
        }
    }
    sift3d_hip_event_record(d->ev[2], d->stream);
    range_stop(rng);
    if (pyramid_only) {
        /* sift3d_amd_build_pyramid_device: the Gaussian pyramid alone (bench.py's pyramid-only leg) */
        for (o = 3; o <= 5; o++)
            sift3d_hip_event_record(d->ev[o], d->stream);      /* (the later stages: empty) */
        if (sift3d_hip_stream_sync(d->stream))
            return SIFT3D_FAILURE;
        d->have_pyramid = 1;
        d->im_valid = im_stored;
        d->last_vol = d_vol == d->d_in ? d_vol : NULL;
        d->ncand = 0;
        d->t_pending |= 1;
        d->t[7] = now_s() - t_start;
        return SIFT3D_SUCCESS;
    }
    rng = range_start("sift3d: DoG maxima");

    /* build_dog (sift.c:713-732) + the dogmax scan (sift.c:821-826).  Default configuration: only
     * the maxima are computed here; the extrema sweep forms the differences itself and no DoG
     * level is stored.  Otherwise (cuboid neighbourhood, another level count, rows that are not
     * whole quads) the octave's DoG levels are stored as the reference does. */
    /* Default configuration on every octave (and a second stream at hand): octave 0 on the main
     * stream, the short launches of octaves >= 1 beside it -- in the DoG stage and again for the
     * extrema sweeps, whose results are then emitted in octave order. */
    if (side && !overlap && (sift3d_hip_event_record(d->ev_fork, d->stream) ||
                             sift3d_hip_stream_wait_event(d->oct_stream, d->ev_fork)))
        return SIFT3D_FAILURE;
    for (o = 0; o < d->num_octaves; o++) {
        const size_t n = (size_t)d->odims[o][0] * d->odims[o][1] * d->odims[o][2];
        int rc = 1;
        d->dog_free[o] = 0;
        if (side && EST_OCTAVE(d, o))
            /* the large octaves (nearly all of the pyramid's bytes): lower bounds of their maxima from a
             * sub-lattice; the extrema sweep gathers the exact ones (sift3d_hip_extrema_gauss6_est_phase) */
            rc = sift3d_hip_dogmax_sub((const float *const *)(d->d_g + o * d->ngl), d->odims[o][0],
                                       d->odims[o][1], d->odims[o][2],
                                       d->d_scalars + 8 + (d->num_octaves + o) * d->ndl,
                                       o > 0 ? d->oct_stream : d->stream);
        else if (!d->cuboid_extrema && d->ngl == 6 && (d->odims[o][0] & 3) == 0 && d->odims[o][2] >= 3)
            rc = sift3d_hip_dogmax_stack((const float *const *)(d->d_g + o * d->ngl), d->ngl, n,
                                         d->d_scalars + 8 + o * d->ndl,
                                         side && o > 0 ? d->oct_stream : d->stream);
        if (rc == SIFT3D_SUCCESS) {
            d->dog_free[o] = 1;
            continue;
        }
        if (side) {
            ERR("sift3d_amd: octave %d is not covered by the DoG-free path \n", o);
            return SIFT3D_FAILURE;
        }
        if (rc != 1 || ensure_dog_octave(d, o))
            return SIFT3D_FAILURE;
        /* one pass over the octave's Gaussian levels when the stack kernel covers it */
        rc = sift3d_hip_dog_stack((const float *const *)(d->d_g + o * d->ngl), d->d_d + o * d->ndl,
                                  d->ngl, n, d->d_scalars + 8 + o * d->ndl, d->stream);
        if (rc == SIFT3D_SUCCESS)
            continue;
        if (rc != 1)
            return SIFT3D_FAILURE;
        for (s = 0; s < d->ndl; s++)
            if (sift3d_hip_subtract_absmax(d->d_g[o * d->ngl + s], d->d_g[o * d->ngl + s + 1],
                                           d->d_d[o * d->ndl + s], n,
                                           d->d_scalars + 8 + o * d->ndl + s, d->stream))
                return SIFT3D_FAILURE;
    }
    /* (overlap: the DoG maxima of octaves >= 1 are on the octave stream; the small octaves' sweeps, on the side
     * stream, wait for them -- the main stream does not) */
    if (side && (sift3d_hip_event_record(d->ev_join, d->oct_stream) ||
                 sift3d_hip_stream_wait_event(overlap ? d->side_stream : d->stream, d->ev_join)))
        return SIFT3D_FAILURE;
    sift3d_hip_event_record(d->ev[3], d->stream);
    range_stop(rng);
    rng = range_start("sift3d: extrema");

    /* detect_extrema, sift.c:735-871 */
    if (d->ndl < 3) {
        printf("detect_extrema: Requires at least 3 levels per octave, provided only %d \n", d->ndl);
        return SIFT3D_FAILURE;
    }
    if (ensure_cand_capacity(d, d->cand_cap ? d->cand_cap : (1u << 18)))
        return SIFT3D_FAILURE;
    for (attempt = 0; attempt < 2; attempt++) {
        int split = 0;
        if (sift3d_hip_memset(d->d_scalars + 1, 0, sizeof(uint32_t), d->stream))
            return SIFT3D_FAILURE;
        if (side) {
            /* the sweeps side by side, then scan + emission in octave order */
            int phase;
            split = overlap && attempt == 0 && d->num_octaves - 1 <= SIFT3D_HIP_EXTREMA_MAX_OCT;
            /* three chains: octave 0 | octaves 1, 2 | the small octaves, whose 4-40 us launches (four per
             * octave, each waiting for its predecessor) otherwise queue behind octave 1's sweep and end the
             * stage 0.1 ms after octave 0 has finished */
            /* (overlap, first attempt: the side chains are already where they must be and do NOT wait for the
             * main stream, which may still be inside octave 0's last blur) */
            if ((!overlap || attempt > 0) &&
                (sift3d_hip_event_record(d->ev_fork, d->stream) ||
                 sift3d_hip_stream_wait_event(d->oct_stream, d->ev_fork) ||
                 sift3d_hip_stream_wait_event(d->side_stream, d->ev_fork)))
                return SIFT3D_FAILURE;
            for (phase = 1; phase <= 2; phase++) {
                if (phase == 2) {
                    /* scan + emission of every octave in two launches (octave order is kept by the scan) */
                    sift3d_hip_extrema_oct oc[32];
                    int rc;
                    for (o = 0; o < d->num_octaves; o++) {
                        oc[o].d_g = (const float *const *)(d->d_g + o * d->ngl);
                        oc[o].nx = d->odims[o][0]; oc[o].ny = d->odims[o][1]; oc[o].nz = d->odims[o][2];
                        oc[o].tag0 = o * d->ngl + 1;
                        oc[o].d_work = o ? (void *)((char *)d->d_work2 + d->work2_off[o]) : d->d_work;
                        oc[o].work_bytes = o ? sift3d_hip_extrema_work_bytes(d->odims[o][0], d->odims[o][1],
                                                                             d->odims[o][2], 3)
                                             : d->work_bytes;
                    }
                    if (split) {
                        /* Octave 0's candidates are the head of the list whatever the smaller octaves hold
                         * (sift.c:835-868: octave order) -- a third of it on the bench volume, whose blobs
                         * put most extrema into octaves >= 1; more where the structure is fine.  Their scan +
                         * emission and their ORIENTATION (device-filling: 1.5 ms for the whole list at 512^3)
                         * start as soon as octave 0's sweep has ended, on the main stream; the chains of the
                         * smaller octaves -- latency-bound launches that end later -- finish beside them, and
                         * their candidates are emitted behind octave 0's and oriented on the octave stream,
                         * whose kernels are dispatched first.  Same list, same order. */
                        uint32_t count_a = 0;
                        volatile uint32_t *h_cnt = (volatile uint32_t *)(d->h_keep + d->cand_cap);
                        /* both emissions are enqueued before the host waits for the first count: the smaller
                         * octaves' scan starts from octave 0's total (ev_part orders the two on the device) and
                         * runs when their sweeps have ended -- not when the host has come back from its wait
                         * and has launched octave 0's orientation kernels, which it would then queue behind */
                        if (sift3d_hip_extrema_gauss6_finish(oc, 1, d->peak_thresh, d->d_cand, d->cand_cap,
                                                             (uint32_t *)(d->d_scalars + 1), d->stream) ||
                            sift3d_hip_memcpy_d2h((void *)(h_cnt + 0), d->d_scalars + 1, sizeof(uint32_t), d->stream) ||
                            sift3d_hip_event_record(d->ev_part, d->stream) ||
                            sift3d_hip_event_record(d->ev_join2, d->side_stream) ||
                            sift3d_hip_stream_wait_event(d->oct_stream, d->ev_join2) ||
                            sift3d_hip_stream_wait_event(d->oct_stream, d->ev_part) ||
                            sift3d_hip_extrema_gauss6_finish(oc + 1, d->num_octaves - 1, d->peak_thresh, d->d_cand,
                                                             d->cand_cap, (uint32_t *)(d->d_scalars + 1),
                                                             d->oct_stream) ||
                            sift3d_hip_memcpy_d2h((void *)(h_cnt + 1), d->d_scalars + 1, sizeof(uint32_t),
                                                  d->oct_stream) ||
                            sift3d_hip_stream_sync(d->stream))
                            return SIFT3D_FAILURE;
                        count = count_a = h_cnt[0];
                        sift3d_hip_event_record(d->ev[4], d->stream);
                        /* (a list that does not fit: the rest is still counted, then everything is grown
                         * below for the second attempt) */
                        if (count_a <= d->cand_cap && orient_scratch(d))
                            return SIFT3D_FAILURE;
                        if (count_a && count_a <= d->cand_cap &&
                            orient_part(d, 0, d->ngl, 0, count_a, 0, d->stream))
                            return SIFT3D_FAILURE;
                        /* (on the main stream, ordinary priority: with the smaller octaves' launches behind
                         * it in the dispatch order the two parts end together -- measured; at the chains'
                         * priority this part ends 0.25 ms earlier and the other one 0.2 ms later) */
                        sift3d_hip_event_record(d->ev_part, d->stream);
                        /* (the records' copy: on the side stream, which has nothing else to do from here on) */
                        if ((count_a && count_a <= d->cand_cap &&
                             sift3d_hip_memcpy_d2h(d->h_cand, d->d_cand, sizeof(sift3d_hip_cand) * (size_t)count_a,
                                                   d->side_stream)) ||
                            sift3d_hip_stream_sync(d->oct_stream))
                            return SIFT3D_FAILURE;
                        count = h_cnt[1];
                        if (count > d->cand_cap)
                            break;              /* (count_a <= count) */
                        if (count > count_a &&
                            (orient_part(d, d->ngl, d->num_octaves * d->ngl, count_a, count - count_a, 1,
                                         d->oct_stream) ||
                             sift3d_hip_memcpy_d2h(d->h_cand + count_a, d->d_cand + count_a,
                                                   sizeof(sift3d_hip_cand) * (size_t)(count - count_a),
                                                   d->oct_stream)))
                            return SIFT3D_FAILURE;
                        if (sift3d_hip_event_record(d->ev_join, d->oct_stream) ||
                            sift3d_hip_stream_wait_event(d->stream, d->ev_join))
                            return SIFT3D_FAILURE;
                        oriented = 1;
                        d->parts_timed = 1;
                        break;
                    }
                    rc = sift3d_hip_extrema_gauss6_finish(oc, d->num_octaves, d->peak_thresh, d->d_cand,
                                                          d->cand_cap, (uint32_t *)(d->d_scalars + 1), d->stream);
                    if (rc == SIFT3D_SUCCESS)
                        break;
                    if (rc != 1)
                        return SIFT3D_FAILURE;
                }
                for (o = 0; o < d->num_octaves; o++) {
                    void *const xs = phase != 1 || o == 0 ? d->stream : o <= 2 ? d->oct_stream : d->side_stream;
                    void *wk = o ? (void *)((char *)d->d_work2 + d->work2_off[o]) : d->d_work;
                    const size_t wb = o ? sift3d_hip_extrema_work_bytes(d->odims[o][0], d->odims[o][1],
                                                                        d->odims[o][2], 3)
                                        : d->work_bytes;
                    if (EST_OCTAVE(d, o)) {
                        if (sift3d_hip_extrema_gauss6_est_phase(
                                (const float *const *)(d->d_g + o * d->ngl),
                                d->d_scalars + 8 + (d->num_octaves + o) * d->ndl,
                                d->d_scalars + 8 + o * d->ndl, d->odims[o][0], d->odims[o][1],
                                d->odims[o][2], o * d->ngl + 1, d->peak_thresh, d->d_cand, d->cand_cap,
                                (uint32_t *)(d->d_scalars + 1), wk, wb, xs, phase))
                            return SIFT3D_FAILURE;
                        continue;
                    }
                    if (sift3d_hip_extrema_gauss6_phase(
                            (const float *const *)(d->d_g + o * d->ngl), d->d_scalars + 8 + o * d->ndl,
                            d->odims[o][0], d->odims[o][1], d->odims[o][2], 1, d->odims[o][2] - 1,
                            o * d->ngl + 1, d->peak_thresh, d->d_cand, d->cand_cap,
                            (uint32_t *)(d->d_scalars + 1), wk, wb, xs, phase))
                        return SIFT3D_FAILURE;   /* (coverage was established by the dogmax calls) */
                }
                if (phase == 1 && !split &&
                    (sift3d_hip_event_record(d->ev_join, d->oct_stream) ||
                     sift3d_hip_stream_wait_event(d->stream, d->ev_join) ||
                     sift3d_hip_event_record(d->ev_join2, d->side_stream) ||
                     sift3d_hip_stream_wait_event(d->stream, d->ev_join2)))
                    return SIFT3D_FAILURE;
            }
        }
        if (split) {
            /* (the counts were read where the two parts were emitted) */
            if (oriented || attempt > 0)
                break;
            /* the list does not fit: nothing of this attempt is kept (its launches must have ended before the
             * arrays they use are replaced) */
            if (sift3d_hip_stream_sync(d->stream) || sift3d_hip_stream_sync(d->oct_stream) ||
                sift3d_hip_stream_sync(d->side_stream) || ensure_cand_capacity(d, count + count / 4 + 1024))
                return SIFT3D_FAILURE;
            continue;
        }
        for (o = 0; o < d->num_octaves && !side; o++) {
            sift3d_hip_extrema_level lv[8];
            const int nl = d->ndl - 2;
            if (nl > 8) {
                ERR("sift3d_amd: at most 8 keypoint levels per octave are supported \n");
                return SIFT3D_FAILURE;
            }
            if (d->dog_free[o]) {
                const int rc = sift3d_hip_extrema_gauss6(
                    (const float *const *)(d->d_g + o * d->ngl), d->d_scalars + 8 + o * d->ndl,
                    d->odims[o][0], d->odims[o][1], d->odims[o][2], 1, d->odims[o][2] - 1,
                    o * d->ngl + 1, d->peak_thresh, d->d_cand, d->cand_cap,
                    (uint32_t *)(d->d_scalars + 1), d->d_work, d->work_bytes, d->stream);
                if (rc == SIFT3D_SUCCESS)
                    continue;
                return SIFT3D_FAILURE;       /* (coverage was established by the dogmax call) */
            }
            for (s = 0; s < nl; s++) {
                lv[s].prev = d->d_d[o * d->ndl + s];
                lv[s].cur = d->d_d[o * d->ndl + s + 1];
                lv[s].next = d->d_d[o * d->ndl + s + 2];
                lv[s].d_absmax = d->d_scalars + 8 + o * d->ndl + s + 1;
                lv[s].z_lo = 1;
                lv[s].z_hi = d->odims[o][2] - 1;
                lv[s].tag = o * d->ngl + s + 1;      /* Gaussian level (o, s) of the table */
            }
            if (sift3d_hip_extrema_mode(lv, nl, d->odims[o][0], d->odims[o][1], d->odims[o][2],
                                   d->peak_thresh, d->cuboid_extrema, d->d_cand, d->cand_cap,
                                   (uint32_t *)(d->d_scalars + 1), d->d_work, d->work_bytes,
                                   d->stream))
                return SIFT3D_FAILURE;
        }
        if (sift3d_hip_memcpy_d2h(&count, d->d_scalars + 1, sizeof(count), d->stream) ||
            sift3d_hip_stream_sync(d->stream))
            return SIFT3D_FAILURE;
        if (count <= d->cand_cap)
            break;
        if (ensure_cand_capacity(d, count + count / 4 + 1024))
            return SIFT3D_FAILURE;
    }
    if (!oriented)
        sift3d_hip_event_record(d->ev[4], d->stream);
    d->ncand = (int)count;
    range_stop(rng);
    rng = range_start("sift3d: orientation");

    /* assign_orientations, sift.c:1109-1167 (the default schedule has started it above, octave 0 first) */
    if (count && !oriented) {
        /* The candidate records are final (the host has just read their count): their copy runs on the
         * side stream beside the orientation kernels. */
        if (orient_scratch(d) ||
            sift3d_hip_memcpy_d2h(d->h_cand, d->d_cand, sizeof(sift3d_hip_cand) * (size_t)count,
                                  d->oct_stream) ||
            orient_part(d, 0, d->num_octaves * d->ngl, 0, count, 0, d->stream))
            return SIFT3D_FAILURE;
    }
    sift3d_hip_event_record(d->ev[5], d->stream);
    if (sift3d_hip_stream_sync(d->stream) || (count && sift3d_hip_stream_sync(d->oct_stream)) ||
        (oriented && sift3d_hip_stream_sync(d->side_stream)))
        return SIFT3D_FAILURE;
    range_stop(rng);

    /* keypoint store: dimensions of the first octave (sift.c:756-759), then the in-place
     * compaction of assign_orientations.  copy_Keypoint does not copy `strength`
     * (sift.c:372-384), so slot j keeps the strength of CANDIDATE j (quirk Q2). */
    kp->nx = d->odims[0][0];
    kp->ny = d->odims[0][1];
    kp->nz = d->odims[0][2];
    d->t[10 + 2 * SIFT3D_AMD_TIMED_BLURS + 1] = now_s();        /* (the compaction's seconds: below) */
    {
        const int nt = host_threads(count);
        size_t pre[HOST_THREADS_MAX + 1];
        int rc = SIFT3D_SUCCESS;
        pre[0] = 0;
#pragma omp parallel num_threads(nt)
        {
            /* the team may be smaller than asked for (a caller inside its own parallel region,
             * OMP_THREAD_LIMIT, OMP_DYNAMIC): the list is cut by the team's real size */
            const int nth = omp_get_num_threads(), t = omp_get_thread_num();
            const size_t lo = (size_t)count * t / nth, hi = (size_t)count * (t + 1) / nth;
            size_t q, jj = 0;
            for (q = lo; q < hi; q++)
                jj += d->h_keep[q] != 0;
            pre[t + 1] = jj;
#pragma omp barrier
#pragma omp single
            {
                int u;
                for (u = 0; u < nth; u++)
                    pre[u + 1] += pre[u];
                rc = kp_store_resize(kp, pre[nth]);
            }
            /* (implicit barrier) */
            if (rc == SIFT3D_SUCCESS) {
                jj = pre[t];
                for (q = lo; q < hi; q++) {
                    const sift3d_hip_cand *c = d->h_cand + q;
                    const sift3d_hip_level *L = d->h_levels + c->tag;
                    keypoint_t *k;
                    uint32_t plane, rem, zq;
                    if (!d->h_keep[q])
                        continue;
                    k = kp->buf + jj;
                    plane = (uint32_t)L->nx * (uint32_t)L->ny;      /* (a level has < 2^32 voxels) */
                    zq = c->idx / plane;
                    rem = c->idx - zq * plane;
                    k->o = c->tag / d->ngl;
                    k->s = c->tag % d->ngl - 1;
                    k->xd = (double)(rem % (uint32_t)L->nx);
                    k->yd = (double)(rem / (uint32_t)L->nx);
                    k->zd = (double)zq;
                    k->sd = L->sd;
                    memcpy(k->R, d->h_R + 9 * q, sizeof(k->R));
                    k->strength = d->h_cand[jj].val;
                    jj++;
                }
            }
        }
        if (rc != SIFT3D_SUCCESS)
            return SIFT3D_FAILURE;
    }
    d->have_pyramid = 1;
    d->im_valid = im_stored;
    /* The scaled image is formed on demand (sift3d_amd_copy_level, which = 2) from the volume -- but only from
     * the detector's OWN upload buffer: a caller's device pointer is not kept beyond the call (it may be
     * freed or reused the moment this returns). */
    d->last_vol = d_vol == d->d_in ? d_vol : NULL;

    d->t_pending |= 1;                  /* (the stage events are read when sift3d_amd_timings asks) */
    d->t[7] = now_s() - t_start;
    d->t[10 + 2 * SIFT3D_AMD_TIMED_BLURS + 1] = now_s() - d->t[10 + 2 * SIFT3D_AMD_TIMED_BLURS + 1];
    return SIFT3D_SUCCESS;
}

int sift3d_amd_detect_keypoints_device(sift3d_detector *d, const float *d_volume, int nx, int ny,
                                       int nz, double ux, double uy, double uz,
                                       sift3d_keypoint_store *store)
{
    if (!d || !d_volume || !store || nx < 1 || ny < 1 || nz < 1)
        return SIFT3D_FAILURE;
    if (!(ux > 0) || !(uy > 0) || !(uz > 0)) {          /* as sift3d_amd_image_set_units */
        ERR("sift3d_amd: voxel spacing must be positive, provided (%f, %f, %f) \n", ux, uy, uz);
        return SIFT3D_FAILURE;
    }
    if (ensure_device(d))
        return SIFT3D_FAILURE;
    return detect_on_device(d, d_volume, nx, ny, nz, ux, uy, uz, store, 0);
}

/* The Gaussian pyramid alone (max|v|, the scaling folded into the first blur, build_gpyr: sift.c:645-649,
 * 662-711) of a volume in device memory -- the part of sift3d_amd_detect_keypoints_device that BASELINE's
 * second metric ("achieved HBM GB/s on the Gauss pyramid") names, for a timed leg of its own: inside a whole
 * step the pyramid's last launches share the device with octave 0's extrema sweep.  Blocks until the levels
 * are complete; sift3d_amd_timings()[1] / [6] hold its device time, sift3d_amd_copy_level reads the levels. */
int sift3d_amd_build_pyramid_device(sift3d_detector *d, const float *d_volume, int nx, int ny, int nz,
                                    double ux, double uy, double uz)
{
    if (!d || !d_volume || nx < 1 || ny < 1 || nz < 1)
        return SIFT3D_FAILURE;
    if (!(ux > 0) || !(uy > 0) || !(uz > 0)) {
        ERR("sift3d_amd: voxel spacing must be positive, provided (%f, %f, %f) \n", ux, uy, uz);
        return SIFT3D_FAILURE;
    }
    if (ensure_device(d))
        return SIFT3D_FAILURE;
    return detect_on_device(d, d_volume, nx, ny, nz, ux, uy, uz, NULL, 1);
}

int sift3d_detect_keypoints(sift3d_detector *const d, const sift3d_image *const im,
                            sift3d_keypoint_store *const kp)
{
    size_t n;
    if (im->nc != 1) {                                 /* sift.c:1221-1226 */
        ERR("SIFT3D_detect_keypoints: invalid number of image channels: %d -- only "
            "single-channel images are supported \n", im->nc);
        return SIFT3D_FAILURE;
    }
    if (!im->data)
        return SIFT3D_FAILURE;                         /* im_copy_data, imutil.c:653-654 */
    if (ensure_device(d))
        return SIFT3D_FAILURE;
    n = (size_t)im->nx * im->ny * im->nz;
    if (n > d->in_cap) {
        sift3d_hip_free(d->d_in);
        d->in_cap = 0;
        if (!(d->d_in = (float *)sift3d_hip_malloc(n * sizeof(float))))
            return SIFT3D_FAILURE;
        d->in_cap = n;
    }
    if (sift3d_hip_memcpy_h2d(d->d_in, im->data, n * sizeof(float), d->stream))
        return SIFT3D_FAILURE;
    return detect_on_device(d, d->d_in, im->nx, im->ny, im->nz, im->ux, im->uy, im->uz, kp, 0);
}

int sift3d_extract_descriptors(sift3d_detector *const d, const sift3d_keypoint_store *const kp,
                               sift3d_descriptor_store *const desc)
{
    const int num = (int)kp->num;
    const double t_start = now_s();
    int i, lvbad = 0, lv_exact = 0;
    size_t n_exact = 0;
    uint64_t rng;

    /* verify_keys, sift.c:1171-1212 (against the retained image dimensions) */
    if (num < 1) {
        ERR("verify_keys: invalid number of keypoints: %d \n", num);
        return SIFT3D_FAILURE;
    }
    {
        /* the checks on a few threads; the first offender (if any) is then reported in list order */
        int bad = 0;
#pragma omp parallel for num_threads(host_threads((size_t)num)) schedule(static) reduction(| : bad, lvbad)
        for (i = 0; i < num; i++) {
            const keypoint_t *k = kp->buf + i;
            const double f = k->o >= 0 && k->o < 64 ? (double)(1ull << k->o) : ldexp(1.0, k->o);
            bad |= k->xd < 0 || k->yd < 0 || k->zd < 0 || k->xd * f >= (double)d->nx ||
                   k->yd * f >= (double)d->ny || k->zd * f >= (double)d->nz || k->sd <= 0;
            lvbad |= k->o < 0 || k->o >= d->num_octaves || k->s < -1 || k->s > d->ngl - 2;
        }
        for (i = 0; i < num && bad; i++) {
            const keypoint_t *k = kp->buf + i;
            const double f = ldexp(1.0, k->o);
            if (k->xd < 0 || k->yd < 0 || k->zd < 0 || k->xd * f >= (double)d->nx ||
                k->yd * f >= (double)d->ny || k->zd * f >= (double)d->nz) {
                ERR("verify_keys: keypoint %d (%f, %f, %f) octave %d exceeds image dimensions "
                    "(%d, %d, %d) \n", i, k->xd, k->yd, k->zd, k->o, d->nx, d->ny, d->nz);
                return SIFT3D_FAILURE;
            }
            if (k->sd <= 0) {
                ERR("verify_keys: keypoint %d has invalid scale %f \n", i, k->sd);
                return SIFT3D_FAILURE;
            }
        }
        if (bad)
            return SIFT3D_FAILURE;       /* (a NaN coordinate: every comparison above is false) */
    }
    /* detector_has_gpyr, sift.c:1544-1549, 1623-1628 */
    if (!d->have_pyramid || !d->num_octaves) {
        ERR("SIFT3D_extract_descriptors: no Gaussian pyramid is available. Make sure "
            "SIFT3D_detect_keypoints was called prior to calling this function. \n");
        return SIFT3D_FAILURE;
    }
    for (i = 0; i < num && lvbad; i++) {
        const keypoint_t *k = kp->buf + i;
        if (k->o < 0 || k->o >= d->num_octaves || k->s < -1 || k->s > d->ngl - 2) {
            ERR("sift3d_amd: keypoint %d refers to pyramid level (%d, %d) which does not exist \n",
                i, k->o, k->s);
            return SIFT3D_FAILURE;
        }
    }
    if ((uint32_t)num > d->kp_cap) {
        const uint32_t cap = (uint32_t)num + (uint32_t)num / 4 + 256;
        sift3d_hip_host_free(d->h_kp);
        d->kp_cap = 0;
        d->h_kp = (sift3d_hip_kp *)sift3d_hip_host_alloc(sizeof(sift3d_hip_kp) * (size_t)cap);
        if (!d->h_kp)
            return SIFT3D_FAILURE;
        d->kp_cap = cap;
    }
    {
        /* Launch order: widest windows first.  The window radius in level voxels grows with
         * the level index s only (14.14 * sigma0 * 2^(s/K)), and a keypoint of the last level
         * costs ~4x one of the first; longest-job-first keeps the tail of the one-wave-per-
         * keypoint kernel short.  row1 sends every histogram to its keypoint's row.
         * A stable counting sort by level on a few threads: per-thread counts per level, then every
         * thread places the keypoints of its part of the list. */
        enum { LV_MAX = 32 };
        const int nt = host_threads((size_t)num), nlv = d->ngl;       /* s + 1 in [0, ngl) */
        size_t cnt[HOST_THREADS_MAX][LV_MAX], start[HOST_THREADS_MAX][LV_MAX];
        if (nlv > LV_MAX) {
            ERR("sift3d_amd: at most %d Gaussian levels per octave are supported \n", LV_MAX);
            return SIFT3D_FAILURE;
        }
        memset(cnt, 0, sizeof(cnt));
        lv_exact = exact_desc_first_level(d->exact_desc, d->ngl, d->num_kp_levels, d->sigma0, d->units);
#pragma omp parallel num_threads(nt)
        {
            const int nth = omp_get_num_threads(), t = omp_get_thread_num();  /* (nth <= nt: see above) */
            const size_t lo = (size_t)num * t / nth, hi = (size_t)num * (t + 1) / nth;
            size_t q;
            for (q = lo; q < hi; q++)
                cnt[t][kp->buf[q].s + 1]++;
#pragma omp barrier
#pragma omp single
            {
                size_t pos = 0;
                int lv, u;
                for (lv = nlv - 1; lv >= 0; lv--) {
                    if (lv == lv_exact - 1)
                        n_exact = pos;         /* (widest windows first: the exact ones lead the list) */
                    for (u = 0; u < nth; u++) {
                        start[u][lv] = pos;
                        pos += cnt[u][lv];
                    }
                }
                if (lv_exact <= 0)
                    n_exact = pos;
            }
            /* (implicit barrier) */
            for (q = lo; q < hi; q++) {
                const keypoint_t *k = kp->buf + q;
                sift3d_hip_kp *r = d->h_kp + start[t][k->s + 1]++;
                memcpy(r->R, k->R, sizeof(r->R));
                r->cx = (float)k->xd;                  /* sift.c:1474-1476 */
                r->cy = (float)k->yd;
                r->cz = (float)k->zd;
                r->level = k->o * d->ngl + k->s + 1;
                r->row1 = (uint32_t)q + 1u;
                r->sd = k->sd;
            }
        }
    }
    /* do_extract_descriptors, sift.c:1561-1596 */
    desc->nx = d->odims[0][0];
    desc->ny = d->odims[0][1];
    desc->nz = d->odims[0][2];
    /* the store's histogram array is page-locked and device-visible */
    if (!desc->pinned || (size_t)num > desc->cap) {
        const size_t cap = (size_t)num + (size_t)num / 8 + 64;
        desc_store_release(desc);
        desc->hist = (float *)sift3d_hip_host_alloc(sizeof(float) * DESC_NUMEL * cap);
        desc->xyzsd = (double *)malloc(sizeof(double) * 4 * cap);
        if (!desc->hist || !desc->xyzsd) {
            desc->pinned = desc->hist != NULL;
            desc_store_release(desc);
            return SIFT3D_FAILURE;
        }
        desc->pinned = 1;
        desc->cap = cap;
    }
    desc->num = (size_t)num;
    desc->d_num = 0;
    if (desc->keep_device && (size_t)num > desc->d_cap) {
        sift3d_hip_free(desc->d_hist);
        desc->d_cap = 0;
        desc->d_hist = (float *)sift3d_hip_malloc(sizeof(float) * DESC_NUMEL * desc->cap);
        if (!desc->d_hist)
            return SIFT3D_FAILURE;
        desc->d_cap = desc->cap;
    }
    d->t_pending &= ~2;
    rng = range_start("sift3d: descriptors");
    sift3d_hip_event_record(d->ev[6], d->stream);
    /* (the kernel reads the 64-byte record of a keypoint once, as its wave starts: straight from the
     * page-locked host list -- no copy, no DMA set-up between the host loops and the launch) */
    /* The kernel stores each histogram straight into the store's page-locked array (mapped
     * into the device's address space): the 3 KB per keypoint trickle over PCIe while the other
     * keypoints are still being computed, so there is no device staging buffer and no D2H
     * copy after the kernel.  (A chunked kernel/copy pipeline measured slower: every chunk
     * pays the kernel's long tail.) */
    {
        float *dev_view = (float *)sift3d_hip_host_device_ptr(desc->hist);
        const sift3d_hip_kp *kp_view = (const sift3d_hip_kp *)sift3d_hip_host_device_ptr(d->h_kp);
        const size_t need = sift3d_hip_describe_part_bytes((uint32_t)(num - (int)n_exact));
        if (need > d->dpart_bytes) {
            sift3d_hip_free(d->d_dpart);
            d->dpart_bytes = 0;
            d->d_dpart = sift3d_hip_malloc(need + need / 8);
            if (!d->d_dpart)
                return SIFT3D_FAILURE;
            d->dpart_bytes = need + need / 8;
        }
        if (!dev_view || !kp_view ||
            sift3d_hip_describe_parts(d->d_levels, d->num_octaves * d->ngl, kp_view, (uint32_t)num,
                                      (uint32_t)n_exact, dev_view, desc->keep_device ? desc->d_hist : NULL,
                                      d->d_wlut, need ? d->d_dpart : NULL, d->stream))
            return SIFT3D_FAILURE;
    }
    sift3d_hip_event_record(d->ev[7], d->stream);
    for (i = 0; i < num; i++) {
        const keypoint_t *k = kp->buf + i;
        const double f = ldexp(1.0, k->o);             /* sift.c:1459, 1530-1533 */
        desc->xyzsd[4 * (size_t)i] = k->xd * f;
        desc->xyzsd[4 * (size_t)i + 1] = k->yd * f;
        desc->xyzsd[4 * (size_t)i + 2] = k->zd * f;
        desc->xyzsd[4 * (size_t)i + 3] = k->sd;
    }
    if (sift3d_hip_stream_sync(d->stream))
        return SIFT3D_FAILURE;
    range_stop(rng);
    if (desc->keep_device)
        desc->d_num = (size_t)num;
    d->t_pending |= 2;
    d->t[8] = now_s() - t_start;
    return SIFT3D_SUCCESS;
}

int sift3d_amd_copy_level(const sift3d_detector *d, int which, int o, int s, float *out, int *dims)
{
    const float *src;
    size_t n;
    if (!d->num_octaves || !d->stream || !d->have_pyramid)
        return SIFT3D_FAILURE;
    if (which == 2) {
        o = 0;
        if (!d->im_valid) {
            /* the scaled image was folded into the first blur: form it now (im_scale) from the volume
             * the detector uploaded itself.  After sift3d_amd_detect_keypoints_device the volume was the
             * CALLER's and no pointer to it was kept: the scaled image is then not available. */
            const size_t n0 = (size_t)d->odims[0][0] * d->odims[0][1] * d->odims[0][2];
            if (!d->last_vol) {
                ERR("sift3d_amd_copy_level: the scaled input image is not retained after "
                    "sift3d_amd_detect_keypoints_device (the volume belongs to the caller) \n");
                return SIFT3D_FAILURE;
            }
            if (sift3d_hip_scale(d->last_vol, d->d_im, n0, d->d_scalars, d->stream))
                return SIFT3D_FAILURE;
        }
        src = d->d_im;
    } else {
        const int nl = which == 0 ? d->ngl : d->ndl;
        if (o < 0 || o >= d->num_octaves || s < -1 || s > nl - 2)
            return SIFT3D_FAILURE;
        if (which == 0) {
            src = d->d_g[o * d->ngl + s + 1];
        } else if (d->d_d[o * d->ndl + s + 1] && !d->dog_free[o]) {
            src = d->d_d[o * d->ndl + s + 1];
        } else {
            /* the DoG level was never stored: form it now (im_subtract, imutil.c:719-739) in
             * scratch */
            const size_t nn = (size_t)d->odims[o][0] * d->odims[o][1] * d->odims[o][2];
            if (sift3d_hip_subtract_absmax(d->d_g[o * d->ngl + s + 1], d->d_g[o * d->ngl + s + 2],
                                           d->d_tmp_a, nn, NULL, d->stream))
                return SIFT3D_FAILURE;
            src = d->d_tmp_a;
        }
    }
    if (dims)
        memcpy(dims, d->odims[o], sizeof(int) * 3);
    if (!out)
        return SIFT3D_SUCCESS;
    n = (size_t)d->odims[o][0] * d->odims[o][1] * d->odims[o][2];
    if (sift3d_hip_memcpy_d2h(out, src, n * sizeof(float), d->stream) ||
        sift3d_hip_stream_sync(d->stream))
        return SIFT3D_FAILURE;
    return SIFT3D_SUCCESS;
}

/* the Z-slab multi-GPU driver (uses the private layouts above) */
#include "sift3d_sharded.c"

/* descriptor matching + RANSAC affine (BASELINE config 5) */
#include "sift3d_register.c"
