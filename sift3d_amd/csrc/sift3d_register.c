/* sift3d_register.c -- descriptor matching and RANSAC affine fit (included at the end of
 * sift3d_host.c).
 *
 * BASELINE config 5.  Upstream SIFT3D "can also perform 3D image registration by matching SIFT3D
 * features and fitting geometric transformations with the RANSAC algorithm" (README-OLD.md:5);
 * this fork REMOVED that code (CHANGES.md:99-103), so there is nothing to cite line by line, no
 * oracle and no fixture: PARITY UNPINNED.  Built from the published description:
 *   matching   nearest / second-nearest neighbour under L2 on the 768-float descriptors
 *              (sift3d_hip_nn2, matrix cores), Lowe's ratio test on the two distances, and the
 *              forward-backward check (a match must be mutual);
 *   fitting    12-parameter affine y = A [x; 1] by least squares inside RANSAC: random minimal
 *              samples of 4 matches, inliers by residual, best consensus refit on its inliers.
 * Validated by recovering a known transform between two synthetic volumes
 * (tests/test_register.py). */

/* A matcher: one stream, scratch that grows on demand and is reused (no allocation per match). */
struct sift3d_amd_matcher {
    void *stream, *ev0, *ev1;
    float *d_work;                     /* sift3d_hip_nn2 scratch (sized for both directions) */
    size_t work_floats;
    float *d_d12;                      /* nearest / second nearest distances, both directions */
    int *d_j;
    float *h_d12;                      /* page-locked mirrors */
    int *h_j;
    size_t res_cap;                    /* capacity of the result arrays, in descriptors (na + nb) */
    float *d_up[2];                    /* upload buffers for stores without a device copy */
    size_t up_cap[2];
    double seconds;                    /* device seconds of the two searches of the last match */
};

int sift3d_amd_descriptor_store_keep_device(sift3d_descriptor_store *d, int on)
{
    if (!d)
        return SIFT3D_FAILURE;
    d->keep_device = on != 0;
    if (!on) {
        sift3d_hip_free(d->d_hist);
        d->d_hist = NULL;
        d->d_cap = d->d_num = 0;
    }
    return SIFT3D_SUCCESS;
}

sift3d_amd_matcher *sift3d_amd_make_matcher(void)
{
    sift3d_amd_matcher *m;
    if (!sift3d_amd_device_available()) {
        ERR("sift3d_amd: no HIP device is available; this library has no CPU path \n");
        return NULL;
    }
    m = (sift3d_amd_matcher *)calloc(1, sizeof(*m));
    if (!m)
        return NULL;
    m->stream = sift3d_hip_stream_create();
    m->ev0 = sift3d_hip_event_create();
    m->ev1 = sift3d_hip_event_create();
    if (!m->stream || !m->ev0 || !m->ev1) {
        sift3d_amd_free_matcher(m);
        return NULL;
    }
    return m;
}

void sift3d_amd_free_matcher(sift3d_amd_matcher *m)
{
    if (!m)
        return;
    if (m->stream)
        sift3d_hip_stream_sync(m->stream);
    sift3d_hip_free(m->d_work); sift3d_hip_free(m->d_d12); sift3d_hip_free(m->d_j);
    sift3d_hip_free(m->d_up[0]); sift3d_hip_free(m->d_up[1]);
    sift3d_hip_host_free(m->h_d12); sift3d_hip_host_free(m->h_j);
    sift3d_hip_event_destroy(m->ev0); sift3d_hip_event_destroy(m->ev1);
    sift3d_hip_stream_destroy(m->stream);
    free(m);
}

double sift3d_amd_matcher_seconds(const sift3d_amd_matcher *m) { return m ? m->seconds : 0.0; }

/* the store's histograms in device memory: its own copy when it is current, else an upload */
static const float *reg_device_hist(sift3d_amd_matcher *m, const sift3d_descriptor_store *d, int which)
{
    const size_t need = DESC_NUMEL * (d->num ? d->num : 1);
    if (d->d_hist && d->d_num == d->num && d->num)
        return d->d_hist;
    if (need > m->up_cap[which]) {
        sift3d_hip_free(m->d_up[which]);
        m->up_cap[which] = 0;
        m->d_up[which] = (float *)sift3d_hip_malloc(sizeof(float) * (need + need / 4));
        if (!m->d_up[which])
            return NULL;
        m->up_cap[which] = need + need / 4;
    }
    if (d->num && sift3d_hip_memcpy_h2d(m->d_up[which], d->hist, sizeof(float) * DESC_NUMEL * d->num, m->stream))
        return NULL;
    return m->d_up[which];
}

/* match_ab[i] = index into b of the match of descriptor i of a, or -1.  nn_thresh: the largest
 * accepted ratio (nearest distance) / (second nearest distance), e.g. 0.8. */
int sift3d_amd_matcher_match(sift3d_amd_matcher *m, const sift3d_descriptor_store *a,
                             const sift3d_descriptor_store *b, double nn_thresh, int *match_ab)
{
    const int na = a ? (int)a->num : 0, nb = b ? (int)b->num : 0;
    const float r2 = (float)(nn_thresh * nn_thresh);
    const float *da, *db;
    int i;
    if (!m || !a || !b || !match_ab || nn_thresh <= 0)
        return SIFT3D_FAILURE;
    for (i = 0; i < na; i++)
        match_ab[i] = -1;
    m->seconds = 0.0;
    if (!na || !nb)
        return SIFT3D_SUCCESS;
    {
        /* one scratch buffer serves both directions: size it for the larger of the two */
        const size_t wf = sift3d_hip_nn2_work_floats(na, nb), wb = sift3d_hip_nn2_work_floats(nb, na);
        const size_t need = wf > wb ? wf : wb, tot = (size_t)na + (size_t)nb;
        if (need > m->work_floats) {
            sift3d_hip_free(m->d_work);
            m->work_floats = 0;
            if (!(m->d_work = (float *)sift3d_hip_malloc(sizeof(float) * (need + need / 4))))
                return SIFT3D_FAILURE;
            m->work_floats = need + need / 4;
        }
        if (tot > m->res_cap) {
            const size_t cap = tot + tot / 4;
            sift3d_hip_free(m->d_d12); sift3d_hip_free(m->d_j);
            sift3d_hip_host_free(m->h_d12); sift3d_hip_host_free(m->h_j);
            m->res_cap = 0;
            m->d_d12 = (float *)sift3d_hip_malloc(sizeof(float) * 2 * cap);
            m->d_j = (int *)sift3d_hip_malloc(sizeof(int) * cap);
            m->h_d12 = (float *)sift3d_hip_host_alloc(sizeof(float) * 2 * cap);
            m->h_j = (int *)sift3d_hip_host_alloc(sizeof(int) * cap);
            if (!m->d_d12 || !m->d_j || !m->h_d12 || !m->h_j)
                return SIFT3D_FAILURE;
            m->res_cap = cap;
        }
    }
    if (!(da = reg_device_hist(m, a, 0)) || !(db = reg_device_hist(m, b, 1)))
        return SIFT3D_FAILURE;
    {
        const size_t tot = (size_t)na + (size_t)nb;
        float *d1 = m->d_d12, *d2 = m->d_d12 + tot;
        /* forward (a -> b) and backward (b -> a) */
        if (sift3d_hip_event_record(m->ev0, m->stream) ||
            sift3d_hip_nn2(da, na, db, nb, DESC_NUMEL, m->d_j, d1, d2, m->d_work, m->stream) ||
            sift3d_hip_nn2(db, nb, da, na, DESC_NUMEL, m->d_j + na, d1 + na, d2 + na, m->d_work, m->stream) ||
            sift3d_hip_event_record(m->ev1, m->stream) ||
            sift3d_hip_memcpy_d2h(m->h_d12, m->d_d12, sizeof(float) * 2 * tot, m->stream) ||
            sift3d_hip_memcpy_d2h(m->h_j, m->d_j, sizeof(int) * tot, m->stream) ||
            sift3d_hip_stream_sync(m->stream))
            return SIFT3D_FAILURE;
        m->seconds = 1e-3 * sift3d_hip_event_elapsed_ms(m->ev0, m->ev1);
        {
            const float *fd1 = m->h_d12, *fd2 = m->h_d12 + tot;
            const int *hj = m->h_j;
            for (i = 0; i < na; i++) {
                const int j = hj[i];
                /* ratio test on squared distances; forward-backward consistency */
                if (j < 0 || !(fd1[i] < r2 * fd2[i]))
                    continue;
                if (hj[na + j] != i || !(fd1[na + j] < r2 * fd2[na + j]))
                    continue;
                match_ab[i] = j;
            }
        }
    }
    return SIFT3D_SUCCESS;
}

/* the same on a matcher made for the call */
int sift3d_amd_nn_match(const sift3d_descriptor_store *a, const sift3d_descriptor_store *b,
                        double nn_thresh, int *match_ab)
{
    sift3d_amd_matcher *m;
    int rc;
    if (!a || !b || !match_ab || nn_thresh <= 0)
        return SIFT3D_FAILURE;
    if (!(m = sift3d_amd_make_matcher()))
        return SIFT3D_FAILURE;
    rc = sift3d_amd_matcher_match(m, a, b, nn_thresh, match_ab);
    sift3d_amd_free_matcher(m);
    return rc;
}

/* coordinates {x, y, z} of descriptor i (octave-0 voxels), e.g. to build the point lists */
int sift3d_amd_descriptor_store_xyz(const sift3d_descriptor_store *d, int i, double *xyz)
{
    if (!d || i < 0 || (size_t)i >= d->num || !xyz)
        return SIFT3D_FAILURE;
    memcpy(xyz, d->xyzsd + 4 * (size_t)i, sizeof(double) * 3);
    return SIFT3D_SUCCESS;
}

/* all coordinates at once: xyz[3 i .. 3 i + 2] = {x, y, z} of descriptor i */
int sift3d_amd_descriptor_store_xyz_all(const sift3d_descriptor_store *d, double *xyz)
{
    size_t i;
    if (!d || (d->num && !xyz))
        return SIFT3D_FAILURE;
    for (i = 0; i < d->num; i++)
        memcpy(xyz + 3 * i, d->xyzsd + 4 * i, sizeof(double) * 3);
    return SIFT3D_SUCCESS;
}

/* solve the 4x4 system M X = R^T for the 3 columns (Gaussian elimination, partial pivoting) */
static int reg_solve4(double M[4][4], double R[3][4], double A[3][4])
{
    double aug[4][7];
    int i, j, k, p;
    for (i = 0; i < 4; i++) {
        for (j = 0; j < 4; j++)
            aug[i][j] = M[i][j];
        for (j = 0; j < 3; j++)
            aug[i][4 + j] = R[j][i];
    }
    for (k = 0; k < 4; k++) {
        double piv;
        p = k;
        for (i = k + 1; i < 4; i++)
            if (fabs(aug[i][k]) > fabs(aug[p][k]))
                p = i;
        if (fabs(aug[p][k]) < 1e-12)
            return SIFT3D_FAILURE;
        if (p != k)
            for (j = 0; j < 7; j++) {
                const double t = aug[k][j];
                aug[k][j] = aug[p][j];
                aug[p][j] = t;
            }
        piv = aug[k][k];
        for (j = k; j < 7; j++)
            aug[k][j] /= piv;
        for (i = 0; i < 4; i++)
            if (i != k) {
                const double f = aug[i][k];
                for (j = k; j < 7; j++)
                    aug[i][j] -= f * aug[k][j];
            }
    }
    for (i = 0; i < 3; i++)
        for (j = 0; j < 4; j++)
            A[i][j] = aug[j][4 + i];
    return SIFT3D_SUCCESS;
}

/* least-squares affine dst = A [src; 1] over the points selected by idx */
/* threads of the RANSAC scoring loop: one per four iterations, at most 16 */
static int reg_threads(int iters)
{
    int t = omp_get_num_procs();
    if (t > 16)
        t = 16;
    if (t > (iters + 3) / 4)
        t = (iters + 3) / 4;
    return t < 1 ? 1 : t;
}

static int reg_fit(const double *src, const double *dst, const int *idx, int m, double A[3][4])
{
    double M[4][4], R[3][4];
    int i, j, k;
    memset(M, 0, sizeof(M));
    memset(R, 0, sizeof(R));
    for (k = 0; k < m; k++) {
        const double *x = src + 3 * (size_t)idx[k], *y = dst + 3 * (size_t)idx[k];
        const double h[4] = { x[0], x[1], x[2], 1.0 };
        for (i = 0; i < 4; i++)
            for (j = 0; j < 4; j++)
                M[i][j] += h[i] * h[j];
        for (i = 0; i < 3; i++)
            for (j = 0; j < 4; j++)
                R[i][j] += y[i] * h[j];
    }
    return reg_solve4(M, R, A);
}

static uint64_t reg_rng(uint64_t *s)
{
    *s ^= *s << 13;
    *s ^= *s >> 7;
    *s ^= *s << 17;
    return *s;
}

/* RANSAC for the affine map dst = A [src; 1] (tform: 3 x 4, row-major) between n point pairs.
 * err_thresh: largest residual (in the units of dst) of an inlier; inlier[i] receives 0 / 1.
 * Deterministic for a given seed.  Fails when fewer than 4 pairs are given or no sample was
 * non-degenerate. */
int sift3d_amd_ransac_affine(const double *src, const double *dst, int n, double err_thresh,
                             int num_iter, uint64_t seed, double *tform, unsigned char *inlier,
                             int *num_inliers)
{
    double best[3][4], A[3][4];
    int *idx, it, i, best_cnt = -1, rc = SIFT3D_FAILURE;
    uint64_t s = seed ? seed : 88172645463325252ull;
    const double thr2 = err_thresh * err_thresh;
    if (!src || !dst || !tform || n < 4 || num_iter < 1 || err_thresh <= 0)
        return SIFT3D_FAILURE;
    idx = (int *)malloc(sizeof(int) * (size_t)n);
    if (!idx)
        return SIFT3D_FAILURE;
    memset(best, 0, sizeof(best));
    /* The samples are drawn first, in sequence (one generator: the result depends on the seed
     * alone); the models are then scored on all host cores, and the first model with the largest
     * consensus wins -- the answer of the sequential loop. */
    {
        int *picks = (int *)malloc(sizeof(int) * 4 * (size_t)num_iter);
        int *cnts = (int *)malloc(sizeof(int) * (size_t)num_iter);
        if (!picks || !cnts) {
            free(picks); free(cnts); free(idx);
            return SIFT3D_FAILURE;
        }
        for (it = 0; it < num_iter; it++) {
            int *pick = picks + 4 * (size_t)it, k, dup;
            for (k = 0; k < 4; k++) {
                do {
                    int q;
                    pick[k] = (int)(reg_rng(&s) % (uint64_t)n);
                    dup = 0;
                    for (q = 0; q < k; q++)
                        dup |= pick[q] == pick[k];
                } while (dup);
            }
        }
        /* (an explicit team: libgomp's default is one thread per CPU it SEES, 256 on a GPU box whose quota is 16
         * cores -- hundreds of spinning threads for 500 short iterations) */
#pragma omp parallel for schedule(dynamic, 4) num_threads(reg_threads(num_iter))
        for (it = 0; it < num_iter; it++) {
            double M[3][4];
            int cnt = 0, j, k;
            if (reg_fit(src, dst, picks + 4 * (size_t)it, 4, M)) {
                cnts[it] = -1;              /* degenerate (coplanar) sample */
                continue;
            }
            for (j = 0; j < n; j++) {
                const double *x = src + 3 * (size_t)j, *y = dst + 3 * (size_t)j;
                double e2 = 0;
                for (k = 0; k < 3; k++) {
                    const double r = M[k][0] * x[0] + M[k][1] * x[1] + M[k][2] * x[2] + M[k][3] - y[k];
                    e2 += r * r;
                }
                cnt += e2 <= thr2;
            }
            cnts[it] = cnt;
        }
        {
            int win = -1;
            for (it = 0; it < num_iter; it++)
                if (cnts[it] > best_cnt) {
                    best_cnt = cnts[it];
                    win = it;
                }
            if (win >= 0 && reg_fit(src, dst, picks + 4 * (size_t)win, 4, A) == 0)
                memcpy(best, A, sizeof(best));
        }
        free(picks);
        free(cnts);
    }
    if (best_cnt >= 4) {
        /* consensus set of the best model, then the least-squares refit on it (twice: the refit
         * may admit a few more points) */
        int pass, m = 0;
        for (pass = 0; pass < 2; pass++) {
            m = 0;
            for (i = 0; i < n; i++) {
                const double *x = src + 3 * (size_t)i, *y = dst + 3 * (size_t)i;
                double e2 = 0;
                int k;
                for (k = 0; k < 3; k++) {
                    const double r = best[k][0] * x[0] + best[k][1] * x[1] + best[k][2] * x[2] + best[k][3] - y[k];
                    e2 += r * r;
                }
                if (e2 <= thr2)
                    idx[m++] = i;
                if (inlier)
                    inlier[i] = e2 <= thr2;
            }
            if (m < 4 || reg_fit(src, dst, idx, m, A))
                break;
            memcpy(best, A, sizeof(best));
        }
        if (m >= 4) {
            memcpy(tform, best, sizeof(best));
            if (num_inliers)
                *num_inliers = m;
            rc = SIFT3D_SUCCESS;
        }
    }
    free(idx);
    return rc;
}
