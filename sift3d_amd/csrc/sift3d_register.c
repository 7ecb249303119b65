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

static int reg_upload(const sift3d_descriptor_store *d, float **dev)
{
    *dev = (float *)sift3d_hip_malloc(sizeof(float) * DESC_NUMEL * (d->num ? d->num : 1));
    if (!*dev)
        return SIFT3D_FAILURE;
    if (d->num && sift3d_hip_memcpy_h2d(*dev, d->hist, sizeof(float) * DESC_NUMEL * d->num, NULL))
        return SIFT3D_FAILURE;
    return SIFT3D_SUCCESS;
}

/* match_ab[i] = index into b of the match of descriptor i of a, or -1.  nn_thresh: the largest
 * accepted ratio (nearest distance) / (second nearest distance), e.g. 0.8. */
int sift3d_amd_nn_match(const sift3d_descriptor_store *a, const sift3d_descriptor_store *b,
                        double nn_thresh, int *match_ab)
{
    const int na = a ? (int)a->num : 0, nb = b ? (int)b->num : 0;
    const float r2 = (float)(nn_thresh * nn_thresh);
    float *da = NULL, *db = NULL, *work = NULL, *d1 = NULL, *d2 = NULL, *h = NULL;
    int *dj = NULL, *hj = NULL, i, rc = SIFT3D_FAILURE;
    if (!a || !b || !match_ab || nn_thresh <= 0)
        return SIFT3D_FAILURE;
    if (!sift3d_amd_device_available()) {
        ERR("sift3d_amd: no HIP device is available; this library has no CPU path \n");
        return SIFT3D_FAILURE;
    }
    for (i = 0; i < na; i++)
        match_ab[i] = -1;
    if (!na || !nb)
        return SIFT3D_SUCCESS;
    {
        /* one scratch buffer serves both directions: size it for the larger of the two */
        const size_t wf = sift3d_hip_nn2_work_floats(na, nb), wb = sift3d_hip_nn2_work_floats(nb, na);
        work = (float *)sift3d_hip_malloc(sizeof(float) * (wf > wb ? wf : wb));
        d1 = (float *)sift3d_hip_malloc(sizeof(float) * 2 * (size_t)(na + nb));
        dj = (int *)sift3d_hip_malloc(sizeof(int) * (size_t)(na + nb));
        h = (float *)malloc(sizeof(float) * 2 * (size_t)(na + nb));
        hj = (int *)malloc(sizeof(int) * (size_t)(na + nb));
    }
    if (!work || !d1 || !dj || !h || !hj || reg_upload(a, &da) || reg_upload(b, &db))
        goto done;
    d2 = d1 + (na + nb);
    /* forward (a -> b) and backward (b -> a) */
    if (sift3d_hip_nn2(da, na, db, nb, DESC_NUMEL, dj, d1, d2, work, NULL) ||
        sift3d_hip_nn2(db, nb, da, na, DESC_NUMEL, dj + na, d1 + na, d2 + na, work, NULL) ||
        sift3d_hip_memcpy_d2h(h, d1, sizeof(float) * 2 * (size_t)(na + nb), NULL) ||
        sift3d_hip_memcpy_d2h(hj, dj, sizeof(int) * (size_t)(na + nb), NULL) ||
        sift3d_hip_stream_sync(NULL))
        goto done;
    {
        const float *fd1 = h, *fd2 = h + (na + nb);
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
    rc = SIFT3D_SUCCESS;
done:
    sift3d_hip_free(da); sift3d_hip_free(db); sift3d_hip_free(work); sift3d_hip_free(d1);
    sift3d_hip_free(dj);
    free(h); free(hj);
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
    for (it = 0; it < num_iter; it++) {
        int pick[4], cnt = 0, k, dup;
        for (k = 0; k < 4; k++) {
            do {
                int q;
                pick[k] = (int)(reg_rng(&s) % (uint64_t)n);
                dup = 0;
                for (q = 0; q < k; q++)
                    dup |= pick[q] == pick[k];
            } while (dup);
        }
        if (reg_fit(src, dst, pick, 4, A))
            continue;                       /* degenerate (coplanar) sample */
        for (i = 0; i < n; i++) {
            const double *x = src + 3 * (size_t)i, *y = dst + 3 * (size_t)i;
            double e2 = 0;
            for (k = 0; k < 3; k++) {
                const double r = A[k][0] * x[0] + A[k][1] * x[1] + A[k][2] * x[2] + A[k][3] - y[k];
                e2 += r * r;
            }
            cnt += e2 <= thr2;
        }
        if (cnt > best_cnt) {
            best_cnt = cnt;
            memcpy(best, A, sizeof(best));
        }
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
