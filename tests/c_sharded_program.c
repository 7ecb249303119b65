/* The Z-slab driver as a C host program sees it (include/sift3d_amd.h): `world` ranks -- here threads of this
 * process over the library's stream-ordered thread transport, in production processes over
 * sift3d_amd_rccl_transport -- each create a slab driver, fill their planes, run the collective detect /
 * describe / descriptor gather, and every rank compares its GLOBAL results with the drop-in single-GPU API
 * (sift3d_detect_keypoints / sift3d_extract_descriptors) run on the whole volume: bit for bit.
 * Compiled by tests/test_c_program.py with gcc -pthread against libsift3d_amd.so only (no HIP headers).
 * usage: c_sharded_program world nx ny nz seed  -> prints "ok <keypoints> <candidates>" or a message, rc != 0 */
#include <pthread.h>
#include <sift3d/imtypes.h>
#include <sift3d/imutil.h>
#include <sift3d/sift.h>
#include <sift3d_amd.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

typedef struct {
    int rank, world, nx, ny, nz;
    sift3d_amd_thread_group *group;
    const float *vol;              /* the whole volume (host) */
    const float *want_kp;          /* N x 3 doubles as the reference's to_mat_rm gives them */
    const double *want_xyz;
    const float *want_desc;        /* N x 771 */
    int want_n, want_cand;
    int rc;
    char msg[256];
} job_t;

static void *run_rank(void *arg)
{
    job_t *J = (job_t *)arg;
    sift3d_amd_transport t;
    sift3d_amd_sharded *S = NULL;
    sift3d_keypoint_store *kp = sift3d_make_keypoint_store();
    sift3d_descriptor_store *own = sift3d_make_descriptor_store(), *all = sift3d_make_descriptor_store();
    sift3d_mat_rm *mk = sift3d_make_mat_rm(), *md = sift3d_make_mat_rm();
    int *own_idx = NULL, n_own = 0, z0, z1, cols, rows;
    J->rc = 1;
    if (!kp || !own || !all || !mk || !md || sift3d_amd_thread_transport(&t, J->group, J->rank)) {
        snprintf(J->msg, sizeof(J->msg), "rank %d: setup failed", J->rank);
        sift3d_amd_thread_group_abort(J->group);
        return NULL;
    }
    S = sift3d_amd_sharded_create(J->nx, J->ny, J->nz, &t, NULL, 1.0, 1.0, 1.0);
    if (!S) {
        snprintf(J->msg, sizeof(J->msg), "rank %d: sift3d_amd_sharded_create refused", J->rank);
        sift3d_amd_thread_group_abort(J->group);
        return NULL;
    }
    sift3d_amd_sharded_own_planes(S, &z0, &z1);
    if (sift3d_hip_memcpy_h2d(sift3d_amd_sharded_input(S), J->vol + (size_t)z0 * J->ny * J->nx,
                              sizeof(float) * (size_t)(z1 - z0) * J->ny * J->nx, NULL) ||
        sift3d_hip_stream_sync(NULL)) {
        snprintf(J->msg, sizeof(J->msg), "rank %d: upload failed", J->rank);
        sift3d_amd_thread_group_abort(J->group);
        goto done;
    }
    if (sift3d_amd_sharded_detect(S, kp)) {
        snprintf(J->msg, sizeof(J->msg), "rank %d: detect failed", J->rank);
        goto done;
    }
    own_idx = (int *)malloc(sizeof(int) * (size_t)(J->want_n + 1));
    if (!own_idx || sift3d_amd_sharded_describe(S, kp, own, own_idx, &n_own))
        n_own = -1;                                  /* (still takes part in the gather: its status word) */
    if (sift3d_amd_sharded_gather_descriptors(S, kp, own, own_idx ? own_idx : &n_own, n_own, all, -1)) {
        snprintf(J->msg, sizeof(J->msg), "rank %d: describe / gather failed", J->rank);
        goto done;
    }
    /* the global keypoint list and the gathered descriptors against the single-GPU API */
    if (sift3d_amd_sharded_num_candidates(S) != J->want_cand ||
        sift3d_keypoint_store_to_mat_rm(kp, mk) || sift3d_descriptor_store_to_mat_rm(all, md)) {
        snprintf(J->msg, sizeof(J->msg), "rank %d: %d candidates (want %d) or to_mat_rm failed", J->rank,
                 sift3d_amd_sharded_num_candidates(S), J->want_cand);
        goto done;
    }
    sift3d_mat_rm_dimensions(mk, &cols, &rows);
    if (rows != J->want_n || cols != 3 ||
        memcmp(sift3d_mat_rm_data(mk), J->want_xyz, sizeof(double) * 3 * (size_t)rows)) {
        snprintf(J->msg, sizeof(J->msg), "rank %d: keypoint matrix differs (%d rows, want %d)", J->rank, rows,
                 J->want_n);
        goto done;
    }
    sift3d_mat_rm_dimensions(md, &cols, &rows);
    if (rows != J->want_n || cols != 771 ||
        memcmp(sift3d_mat_rm_data(md), J->want_desc, sizeof(float) * 771 * (size_t)rows)) {
        snprintf(J->msg, sizeof(J->msg), "rank %d: gathered descriptors differ", J->rank);
        goto done;
    }
    J->rc = 0;
done:
    free(own_idx);
    sift3d_amd_sharded_free(S);
    sift3d_amd_thread_transport_free(&t);
    sift3d_free_keypoint_store(kp);
    sift3d_free_descriptor_store(own);
    sift3d_free_descriptor_store(all);
    sift3d_free_mat_rm(mk);
    sift3d_free_mat_rm(md);
    return NULL;
}

int main(int argc, char **argv)
{
    const int world = argc > 1 ? atoi(argv[1]) : 2, nx = argc > 2 ? atoi(argv[2]) : 64,
              ny = argc > 3 ? atoi(argv[3]) : 64, nz = argc > 4 ? atoi(argv[4]) : 256;
    unsigned long long s = argc > 5 ? strtoull(argv[5], NULL, 10) : 5ull;
    const size_t n = (size_t)nx * ny * nz;
    sift3d_image *image = sift3d_make_image(nx, ny, nz, 1);
    sift3d_detector *det = sift3d_make_detector();
    sift3d_keypoint_store *kp = sift3d_make_keypoint_store();
    sift3d_descriptor_store *de = sift3d_make_descriptor_store();
    sift3d_mat_rm *mk = sift3d_make_mat_rm(), *md = sift3d_make_mat_rm();
    sift3d_amd_thread_group *group;
    pthread_t th[16];
    job_t job[16];
    float *data;
    int r, cols, rows, bad = 0;
    if (world < 1 || world > 16 || !image || !det || !kp || !de || !mk || !md)
        return 2;
    data = sift3d_image_data(image);
    for (size_t k = 0; k < n; k++) {                 /* xorshift noise + smooth bumps (many keypoints) */
        s ^= s << 13; s ^= s >> 7; s ^= s << 17;
        data[k] = (float)((s >> 11) * (1.0 / 9007199254740992.0)) * 0.05f;
    }
    for (r = 0; r < 400; r++) {
        int cx, cy, cz, dx, dy, dz;
        s ^= s << 13; s ^= s >> 7; s ^= s << 17;
        cx = 4 + (int)(s % (unsigned)(nx - 8));
        cy = 4 + (int)((s >> 20) % (unsigned)(ny - 8));
        cz = 4 + (int)((s >> 40) % (unsigned)(nz - 8));
        for (dz = -3; dz <= 3; dz++)
            for (dy = -3; dy <= 3; dy++)
                for (dx = -3; dx <= 3; dx++)
                    data[(size_t)(cx + dx) + (size_t)nx * ((size_t)(cy + dy) + (size_t)ny * (cz + dz))] +=
                        (r & 1 ? 1.0f : -1.0f) / (1.0f + (float)(dx * dx + dy * dy + dz * dz));
    }
    if (sift3d_detect_keypoints(det, image, kp) || sift3d_extract_descriptors(det, kp, de) ||
        sift3d_keypoint_store_to_mat_rm(kp, mk) || sift3d_descriptor_store_to_mat_rm(de, md)) {
        fprintf(stderr, "single-GPU reference run failed\n");
        return 3;
    }
    sift3d_mat_rm_dimensions(md, &cols, &rows);
    group = sift3d_amd_thread_group_create(world);
    if (!group)
        return 4;
    for (r = 0; r < world; r++) {
        memset(&job[r], 0, sizeof(job[r]));
        job[r].rank = r; job[r].world = world; job[r].nx = nx; job[r].ny = ny; job[r].nz = nz;
        job[r].group = group; job[r].vol = data;
        job[r].want_xyz = (const double *)sift3d_mat_rm_data(mk);
        job[r].want_desc = (const float *)sift3d_mat_rm_data(md);
        job[r].want_n = rows;
        job[r].want_cand = sift3d_amd_num_candidates(det);
        if (pthread_create(&th[r], NULL, run_rank, &job[r]))
            return 5;
    }
    for (r = 0; r < world; r++) {
        pthread_join(th[r], NULL);
        if (job[r].rc) {
            fprintf(stderr, "%s\n", job[r].msg);
            bad = 1;
        }
    }
    sift3d_amd_thread_group_free(group);
    if (!bad)
        printf("ok %d %d\n", rows, sift3d_amd_num_candidates(det));
    sift3d_free_image(image); sift3d_free_detector(det);
    sift3d_free_keypoint_store(kp); sift3d_free_descriptor_store(de);
    sift3d_free_mat_rm(mk); sift3d_free_mat_rm(md);
    return bad ? 1 : 0;
}
